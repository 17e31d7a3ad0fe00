// Placement of a host-cores rollout's threads: see host_placement.hip.
#pragma once
#include <string>
#include <vector>

#include "coevo_common.hip.h"

namespace coevo {

struct HostTopology {
    std::string allowed;                  // the calling thread's affinity mask as a cpulist
    std::vector<std::string> node_cpus;   // per NUMA node: /sys/devices/system/node/node<N>/cpulist ("" = no such node)
    std::string l3_groups;                // ';'-separated: the distinct cache/index3/shared_cpu_list of the allowed CPUs
    std::string smt_groups;               // ';'-separated: the distinct topology/thread_siblings_list
};

std::vector<int> parse_cpulist(const char *s);
// -> number of CPUs chosen (<= n_threads; 0: nothing to pin to); cpus[0] is the caller's
int choose_placement(const char *allowed, const char *node_cpus, const char *l3_groups, const char *smt_groups, int caller_cpu,
                     int n_threads, int ctx_index, std::vector<int> &cpus, int &flags);
void probe_topology(HostTopology &t);
int node_of_cpu(const HostTopology &t, int cpu);
int numa_node_of_pci(const char *bdf);   // /sys/bus/pci/devices/<bdf>/numa_node, -1 when unknown

}  // namespace coevo
