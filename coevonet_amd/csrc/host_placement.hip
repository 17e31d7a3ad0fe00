// Where the host cores of an env-on-the-host rollout run (north_star: "vectorised env stepping runs on the host cores").
//
// The reference steps its env on whatever core runs the interpreter (utils/game_logic_functions.py:138,179-190).  On the
// two-socket host of a GPU box that choice is worth a factor of four in the step itself: a core on the socket that does NOT
// hold the GPU (and the page-locked observation / action buffers the GPU reads and writes) stepped 752 games in 45-49 us,
// a core next to them in 11 us (profiles/r04_experiments.md section 1, BENCH_r04 cfg2_host_env: 44.8 us).  So a context
//   - asks which NUMA node the GPU hangs off (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<bdf>/numa_node),
//   - intersects that node's CPUs with the thread's affinity mask,
//   - takes ONE L3 complex of it (distinct physical cores first, SMT siblings only when cores run out; T > complex size:
//     the neighbouring complexes), offset by a context index so that two ranks / two contexts do not share cores,
//   - pins the caller's thread to the first of those CPUs for the duration of a rollout (mask saved / restored) and the
//     workers to the others, and allocates the page-locked buffers from that CPU (first touch = that node).
// The choice itself, coevo_host_placement_choose, is a pure function of strings in sysfs cpulist syntax, so that it is tested
// without a GPU and without the topology (tests/test_host_logic_cpu.py); coevo_host_placement_probe reads the strings.
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_placement.hip.h"

namespace coevo {

// "0-7,128-135" -> sorted unique CPU numbers; anything unparsable ends the list (an empty string is an empty set)
std::vector<int> parse_cpulist(const char *s)
{
    std::vector<int> out;
    if (!s) return out;
    const char *p = s;
    while (*p) {
        while (*p == ' ' || *p == ',' || *p == '\n' || *p == '\t') ++p;
        if (!isdigit((unsigned char)*p)) break;
        char *e = nullptr;
        long a = strtol(p, &e, 10), b = a;
        p = e;
        if (*p == '-') {
            ++p;
            if (!isdigit((unsigned char)*p)) break;
            b = strtol(p, &e, 10);
            p = e;
        }
        if (a < 0 || b < a || b > 65535) break;
        for (long c = a; c <= b; ++c) out.push_back((int)c);
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return out;
}

// ';'-separated cpulists -> one set per group
static std::vector<std::vector<int>> parse_groups(const char *s)
{
    std::vector<std::vector<int>> out;
    if (!s) return out;
    std::string cur;
    for (const char *p = s;; ++p) {
        if (*p == ';' || *p == '\0') {
            std::vector<int> g = parse_cpulist(cur.c_str());
            if (!g.empty()) out.push_back(g);
            cur.clear();
            if (!*p) break;
        } else {
            cur.push_back(*p);
        }
    }
    return out;
}

static bool contains(const std::vector<int> &v, int c) { return std::binary_search(v.begin(), v.end(), c); }

static std::vector<int> intersect(const std::vector<int> &a, const std::vector<int> &b)
{
    std::vector<int> out;
    std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(out));
    return out;
}

int choose_placement(const char *allowed_s, const char *node_s, const char *l3_s, const char *smt_s, int caller_cpu,
                     int n_threads, int ctx_index, std::vector<int> &cpus, int &flags)
{
    cpus.clear();
    flags = 0;
    if (n_threads < 1) return 0;
    const std::vector<int> allowed = parse_cpulist(allowed_s);
    if (allowed.empty()) return 0;
    const std::vector<int> node = parse_cpulist(node_s);
    std::vector<int> cand = allowed;
    if (node.empty()) {
        flags |= COEVO_PLACE_NODE_UNKNOWN;
    } else {
        const std::vector<int> on = intersect(allowed, node);
        if (!on.empty()) {
            cand = on;
            flags |= COEVO_PLACE_ON_NODE;
        }   // else: no core of the wanted node is allowed - the flag stays clear and the caller reports it
    }
    // the L3 complexes, restricted to the candidates, in order of their lowest CPU; CPUs no group names form a last group
    std::vector<std::vector<int>> groups;
    {
        std::vector<int> named;
        for (const auto &g : parse_groups(l3_s)) {
            std::vector<int> in = intersect(g, cand);
            std::vector<int> fresh;
            for (int c : in)
                if (!contains(named, c)) fresh.push_back(c);
            if (fresh.empty()) continue;
            groups.push_back(fresh);
            named.insert(named.end(), fresh.begin(), fresh.end());
            std::sort(named.begin(), named.end());
        }
        std::vector<int> rest;
        for (int c : cand)
            if (!contains(named, c)) rest.push_back(c);
        if (!rest.empty()) groups.push_back(rest);
        std::sort(groups.begin(), groups.end(), [](const std::vector<int> &a, const std::vector<int> &b) { return a[0] < b[0]; });
    }
    // distinct physical cores first: a CPU's core is named by the lowest member of its SMT sibling set
    const std::vector<std::vector<int>> smt = parse_groups(smt_s);
    auto core_of = [&](int c) {
        for (const auto &g : smt)
            if (contains(g, c)) return g[0];
        return c;
    };
    const int G = (int)groups.size();
    int start = 0;
    for (int g = 0; g < G; ++g)
        if (contains(groups[g], caller_cpu)) start = g;   // the caller's own complex when it is a candidate: no migration
    if (ctx_index < 0) ctx_index = 0;
    start = (start + ctx_index) % G;
    int first_group = -1;
    bool one_l3 = true;
    std::vector<int> used_cores;
    auto take = [&](int c, int g) {
        cpus.push_back(c);
        used_cores.push_back(core_of(c));
        if (first_group < 0) first_group = g;
        else if (g != first_group) one_l3 = false;
    };
    if (contains(groups[start], caller_cpu)) take(caller_cpu, start);   // the caller keeps the CPU it runs on (slot 0)
    for (int pass = 0; pass < 2 && (int)cpus.size() < n_threads; ++pass)
        for (int i = 0; i < G && (int)cpus.size() < n_threads; ++i) {
            const int g = (start + i) % G;
            for (int c : groups[g]) {
                if ((int)cpus.size() >= n_threads) break;
                if (std::find(cpus.begin(), cpus.end(), c) != cpus.end()) continue;
                const bool core_used = std::find(used_cores.begin(), used_cores.end(), core_of(c)) != used_cores.end();
                if (pass == 0 && core_used) continue;   // SMT siblings of a taken core only when the cores have run out
                take(c, g);
            }
        }
    if (cpus.empty()) return 0;
    if (one_l3) flags |= COEVO_PLACE_ONE_L3;
    if ((int)cpus.size() < n_threads) flags |= COEVO_PLACE_SHORT;   // fewer CPUs than threads: the last threads stay unpinned
    return (int)cpus.size();
}

static std::string read_line(const std::string &path)
{
    std::string out;
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return out;
    char buf[4096];
    if (fgets(buf, sizeof(buf), f)) out = buf;
    fclose(f);
    while (!out.empty() && (out.back() == '\n' || out.back() == ' ')) out.pop_back();
    return out;
}

static std::string cpulist_of_mask(const cpu_set_t &set)
{
    std::string out;
    for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &set)) {
            int e = c;
            while (e + 1 < CPU_SETSIZE && CPU_ISSET(e + 1, &set)) ++e;
            if (!out.empty()) out += ",";
            out += std::to_string(c);
            if (e > c) out += "-" + std::to_string(e);
            c = e;
        }
    return out;
}

void probe_topology(HostTopology &t)
{
    t = HostTopology();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) t.allowed = cpulist_of_mask(set);
    const std::vector<int> allowed = parse_cpulist(t.allowed.c_str());
    for (int n = 0; n < 64; ++n) {
        const std::string s = read_line("/sys/devices/system/node/node" + std::to_string(n) + "/cpulist");
        if (s.empty() && n > 0 && (int)t.node_cpus.size() <= n - 8) break;   // (node numbers may have gaps: look a little further)
        if ((int)t.node_cpus.size() <= n) t.node_cpus.resize((size_t)n + 1);
        t.node_cpus[(size_t)n] = s;
    }
    while (!t.node_cpus.empty() && t.node_cpus.back().empty()) t.node_cpus.pop_back();
    std::vector<std::string> l3_seen, smt_seen;
    for (int c : allowed) {
        const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string(c);
        const std::string l3 = read_line(base + "/cache/index3/shared_cpu_list");
        if (!l3.empty() && std::find(l3_seen.begin(), l3_seen.end(), l3) == l3_seen.end()) l3_seen.push_back(l3);
        const std::string sib = read_line(base + "/topology/thread_siblings_list");
        if (!sib.empty() && std::find(smt_seen.begin(), smt_seen.end(), sib) == smt_seen.end()) smt_seen.push_back(sib);
    }
    for (const auto &s : l3_seen) t.l3_groups += (t.l3_groups.empty() ? "" : ";") + s;
    for (const auto &s : smt_seen) t.smt_groups += (t.smt_groups.empty() ? "" : ";") + s;
}

int node_of_cpu(const HostTopology &t, int cpu)
{
    for (size_t n = 0; n < t.node_cpus.size(); ++n)
        if (contains(parse_cpulist(t.node_cpus[n].c_str()), cpu)) return (int)n;
    return -1;
}

int numa_node_of_pci(const char *bdf)
{
    if (!bdf || !*bdf) return -1;
    std::string b(bdf);
    for (auto &ch : b) ch = (char)tolower((unsigned char)ch);
    const std::string s = read_line("/sys/bus/pci/devices/" + b + "/numa_node");
    if (s.empty()) return -1;
    char *e = nullptr;
    const long v = strtol(s.c_str(), &e, 10);
    return (e == s.c_str() || v < 0) ? -1 : (int)v;
}

}  // namespace coevo

extern "C" int coevo_host_placement_choose(const char *allowed, const char *node_cpus, const char *l3_groups,
                                           const char *smt_groups, int caller_cpu, int n_threads, int ctx_index,
                                           int32_t *cpus_out, int32_t *flags_out)
{
    if (!allowed || !cpus_out || n_threads < 1 || n_threads > 256) return COEVO_ERR_ARG;
    std::vector<int> cpus;
    int flags = 0;
    const int n = coevo::choose_placement(allowed, node_cpus, l3_groups, smt_groups, caller_cpu, n_threads, ctx_index, cpus, flags);
    for (int i = 0; i < n; ++i) cpus_out[i] = cpus[(size_t)i];
    if (flags_out) *flags_out = flags;
    return n;
}

COEVO_DEFINE_TU_FLAGS(host_placement)
