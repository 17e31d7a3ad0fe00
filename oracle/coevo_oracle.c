/*
 * coevo_oracle.c  --  TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, sequential) of the reference's fitness/rollout hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product path
 * (coevonet_amd/) never does and fails loudly when its HIP library is missing.
 *
 * Parity status: PINNED.  oracle/ref_port.py drives these functions in the reference's call order
 * and tests/test_oracle_golden.py checks the results against the JSON fixtures in tests/golden/, which were minted
 * by tests/golden/make_golden.py from the reference's own Python run in the build container.
 *
 * What each function follows (file:line under /root/reference):
 *   oracle_fc_forward        MPE/fcnetwork.py:37-90   (forward + first-max determine_action)
 *   oracle_play_game         utils/game_logic_functions.py:123-212, :215-228  (play_MPE / play_game)
 *   oracle_mpe_*             PettingZoo MPE simple_adversary (third-party, un-vendored, un-pinned:
 *                            requirements.txt:5); published semantics restated, see
 *                            coevonet_amd/mpe/simple_adversary.py header.  Env fidelity vs PettingZoo
 *                            itself is UNPINNED (no PettingZoo in the image); the reference loop was
 *                            run on this same env to mint the fixtures, so the hot path is pinned.
 *   oracle_diversity         utils/game_logic_functions.py:12-37
 *   oracle_perturb_philox    agent.py:25-29 (GA) / :51-53 (ES) with the build's counter-based noise
 *                            (device_philox mode; not a reference RNG stream)
 *   oracle_es_update_from_pert  evolutionary_strategy.py:120-148
 *   oracle_deepqn_forward    Atari/deepqn.py:39-48
 *
 * CANONICAL fp32 ARITHMETIC (the HIP kernels reproduce exactly this, so HIP == oracle bit for bit;
 * the reference's torch-CPU BLAS order is unknowable, so oracle vs reference is a tolerance check on
 * logits and an exact check on actions/rewards wherever the top-2 logit margin is safe):
 *   Linear:     acc = bias[j]; for k = 0..K-1: acc = fmaf(W[j][k], x[k], acc)
 *   Reduce(N):  N values in blocks of 64 consecutive indices; inside a block a balanced binary tree,
 *               adjacent pairs first (lane xor 1, 2, 4, 8, 16, 32); block sums added left to right
 *   LayerNorm:  mean = Reduce(x) * (1/N); d = x - mean; var = Reduce(d*d) * (1/N);
 *               rstd = 1.0f / sqrtf(var + 1e-5f); y = fmaf(d * rstd, gamma, beta)
 *   ReLU:       y > 0 ? y : 0
 *   argmax:     strict '>' scan from -inf, first maximum (fcnetwork.py:78-85)
 * Build: gcc -O2 -ffp-contract=off -mfma (see oracle/Makefile); fmaf must be a true fused op.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define H1 512
#define H2 256
#define NACT 5
#define LN_EPS 1e-5f

enum {
    ST_BAD_INPUT = 1, ST_BAD_FC1 = 2, ST_BAD_FC2 = 4, ST_BAD_OUT = 8, ST_NO_ACTION = 16
};

/* ------------------------------------------------------------------ canonical reductions */
static float block_tree64(const float *v, int n) /* n <= 64, power of two */
{
    float t[64];
    memcpy(t, v, sizeof(float) * n);
    for (int w = 1; w < n; w <<= 1)
        for (int i = 0; i < n; i += 2 * w)
            t[i] = t[i] + t[i + w];
    return t[0];
}

static float reduce_canon(const float *v, int n) /* n multiple of 64 */
{
    float s = block_tree64(v, 64);
    for (int b = 1; b < n / 64; ++b)
        s = s + block_tree64(v + 64 * b, 64);
    return s;
}

static int bad_post_relu(float y) { return isnan(y) || (isinf(y) && y > 0); }

static void layernorm_relu(float *x, int n, const float *gamma, const float *beta, int *bad)
{
    float tmp[H1];
    float inv_n = 1.0f / (float)n;
    float mean = reduce_canon(x, n) * inv_n;
    for (int j = 0; j < n; ++j) {
        float d = x[j] - mean;
        x[j] = d;
        tmp[j] = d * d;
    }
    float var = reduce_canon(tmp, n) * inv_n;
    float rstd = 1.0f / sqrtf(var + LN_EPS);
    for (int j = 0; j < n; ++j) {
        float y = fmaf(x[j] * rstd, gamma[j], beta[j]);
        if (bad_post_relu(y)) *bad = 1;
        x[j] = (y > 0.0f) ? y : (isnan(y) ? y : 0.0f);
    }
}

static void linear_seq(const float *W, const float *b, const float *x, float *y, int n_out, int n_in)
{
    for (int j = 0; j < n_out; ++j) {
        float acc = b[j];
        const float *w = W + (size_t)j * n_in;
        for (int k = 0; k < n_in; ++k)
            acc = fmaf(w[k], x[k], acc);
        y[j] = acc;
    }
}

/* flat parameter order = torch parameters() order of FCNetwork (MPE/fcnetwork.py:14-22):
 * fc1.w[512*D] fc1.b[512] ln1.w[512] ln1.b[512] fc2.w[256*512] fc2.b[256] ln2.w ln2.b out.w[5*256] out.b[5] */
int oracle_fc_param_count(int D) { return H1 * D + 3 * H1 + H2 * H1 + 3 * H2 + NACT * H2 + NACT; }

int oracle_fc_forward(const float *p, int D, const float *obs, float *logits, int *status)
{
    const float *W1 = p, *b1 = W1 + H1 * D, *g1 = b1 + H1, *be1 = g1 + H1;
    const float *W2 = be1 + H1, *b2 = W2 + H2 * H1, *g2 = b2 + H2, *be2 = g2 + H2;
    const float *W3 = be2 + H2, *b3 = W3 + NACT * H2;
    float h1[H1], h2[H2];
    int st = 0, bad = 0;
    for (int k = 0; k < D; ++k)
        if (!isfinite(obs[k])) st |= ST_BAD_INPUT;
    linear_seq(W1, b1, obs, h1, H1, D);
    layernorm_relu(h1, H1, g1, be1, &bad);
    if (bad) st |= ST_BAD_FC1;
    bad = 0;
    linear_seq(W2, b2, h1, h2, H2, H1);
    layernorm_relu(h2, H2, g2, be2, &bad);
    if (bad) st |= ST_BAD_FC2;
    linear_seq(W3, b3, h2, logits, NACT, H2);
    int best = -1;
    float cur = -INFINITY;
    for (int i = 0; i < NACT; ++i) {
        if (!isfinite(logits[i])) st |= ST_BAD_OUT;
        if (logits[i] > cur) { cur = logits[i]; best = i; }
    }
    if (best < 0) st |= ST_NO_ACTION;
    if (status) *status |= st;
    return best;
}

/* ------------------------------------------------------------------ PCG64 (numpy's bit generator) */
typedef unsigned __int128 u128;
static const u128 PCG_MULT = (((u128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL;

typedef struct { u128 state, inc; } pcg64_t;

static uint64_t pcg_out(u128 s)
{
    uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
    uint64_t x = hi ^ lo;
    unsigned r = (unsigned)(hi >> 58);
    return (x >> r) | (x << ((64 - r) & 63));
}

static uint64_t pcg_next(pcg64_t *g)
{
    g->state = g->state * PCG_MULT + g->inc;
    return pcg_out(g->state);
}

static void pcg_advance(pcg64_t *g, u128 delta)
{
    u128 acc_mult = 1, acc_plus = 0, cur_mult = PCG_MULT, cur_plus = g->inc;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    g->state = acc_mult * g->state + acc_plus;
}

/* ------------------------------------------------------------------ MPE simple_adversary (fp64) */
typedef struct {
    double ppos[3][2], pvel[3][2], lm[2][2];
    int goal;
} mpe_state;

#define MPE_DT 0.1
#define MPE_DAMP 0.25
static int g_pos_first = 1; /* coevonet_amd.mpe.simple_adversary.INTEGRATE_POS_FIRST */
void oracle_mpe_set_pos_first(int v) { g_pos_first = v; }

/* Reset number `ordinal` of the seeded stream (ordinal 0 = the reset inside initialize_env,
 * utils/game_logic_functions.py:54).  One reset = Generator.choice(2) (a buffered 32-bit draw, Lemire
 * range 2 => top bit) + 10 doubles; two resets consume 21 raw 64-bit outputs. */
void oracle_mpe_reset(uint64_t st_hi, uint64_t st_lo, uint64_t inc_hi, uint64_t inc_lo,
                      uint64_t ordinal, mpe_state *s)
{
    pcg64_t g;
    g.state = ((u128)st_hi << 64) | st_lo;
    g.inc = ((u128)inc_hi << 64) | inc_lo;
    uint64_t pair = ordinal >> 1;
    pcg_advance(&g, (u128)pair * 21);
    uint64_t c = pcg_next(&g);
    uint32_t half = (ordinal & 1) ? (uint32_t)(c >> 32) : (uint32_t)c;
    s->goal = (int)(((uint64_t)half * 2) >> 32);
    if (ordinal & 1) pcg_advance(&g, 10);
    double d[10];
    for (int i = 0; i < 10; ++i)
        d[i] = -1.0 + 2.0 * ((double)(pcg_next(&g) >> 11) * (1.0 / 9007199254740992.0));
    for (int a = 0; a < 3; ++a) {
        s->ppos[a][0] = d[2 * a];
        s->ppos[a][1] = d[2 * a + 1];
        s->pvel[a][0] = s->pvel[a][1] = 0.0;
    }
    for (int l = 0; l < 2; ++l) {
        s->lm[l][0] = d[6 + 2 * l];
        s->lm[l][1] = d[7 + 2 * l];
    }
}

void oracle_mpe_observe(const mpe_state *s, int slot, float *obs)
{
    const double *me = s->ppos[slot];
    int n = 0;
    if (slot != 0) {
        obs[n++] = (float)(s->lm[s->goal][0] - me[0]);
        obs[n++] = (float)(s->lm[s->goal][1] - me[1]);
    }
    for (int l = 0; l < 2; ++l) {
        obs[n++] = (float)(s->lm[l][0] - me[0]);
        obs[n++] = (float)(s->lm[l][1] - me[1]);
    }
    for (int j = 0; j < 3; ++j) {
        if (j == slot) continue;
        obs[n++] = (float)(s->ppos[j][0] - me[0]);
        obs[n++] = (float)(s->ppos[j][1] - me[1]);
    }
}

void oracle_mpe_world_step(mpe_state *s, const int *act, double *r_good, double *r_adv)
{
    for (int i = 0; i < 3; ++i) {
        double u[2] = {0.0, 0.0};
        if (act[i] == 1) u[0] = -1.0;
        if (act[i] == 2) u[0] = +1.0;
        if (act[i] == 3) u[1] = -1.0;
        if (act[i] == 4) u[1] = +1.0;
        for (int c = 0; c < 2; ++c) {
            double f = (((u[c] * 5.0) + 0.0) / 1.0) * MPE_DT;
            if (g_pos_first) s->ppos[i][c] = s->ppos[i][c] + s->pvel[i][c] * MPE_DT;
            s->pvel[i][c] = s->pvel[i][c] * (1 - MPE_DAMP);
            s->pvel[i][c] = s->pvel[i][c] + f;
            if (!g_pos_first) s->ppos[i][c] = s->ppos[i][c] + s->pvel[i][c] * MPE_DT;
        }
    }
    double d[3];
    for (int i = 0; i < 3; ++i) {
        double dx = s->ppos[i][0] - s->lm[s->goal][0];
        double dy = s->ppos[i][1] - s->lm[s->goal][1];
        d[i] = sqrt(dx * dx + dy * dy);
    }
    *r_adv = -d[0];
    double m = (d[2] < d[1]) ? d[2] : d[1];
    *r_good = -m + d[0];
}

/* One game, literally the AEC bookkeeping the reference loop sees (quirk Q1 included):
 * nets[slot] = flat params of the net acting for env slot 0 adversary_0 / 1 agent_0 / 2 agent_1.
 * limit < 0 = None.  Returns rewards in play_game's order (agent_0, agent_1, adversary_0). */
int oracle_play_game(const float *net_adv, const float *net_a0, const float *net_a1,
                     uint64_t st_hi, uint64_t st_lo, uint64_t inc_hi, uint64_t inc_lo, uint64_t ordinal,
                     int limit, int max_cycles, double *rewards_out, int *actions_out,
                     float *min_margin_out, int *status_out)
{
    const float *nets[3] = {net_adv, net_a0, net_a1};
    static const int D[3] = {8, 10, 10};
    mpe_state s;
    oracle_mpe_reset(st_hi, st_lo, inc_hi, inc_lo, ordinal, &s);
    double cum[3] = {0, 0, 0}, rew[3] = {0, 0, 0}, acc[3] = {0, 0, 0};
    int act[3] = {0, 0, 0};
    int sel = 0, world_steps = 0, trunc = 0, timesteps = 0, status = 0;
    float min_margin = INFINITY;
    for (;;) {
        int agent = sel;
        float obs[10], logits[NACT];
        oracle_mpe_observe(&s, agent, obs);
        int a = oracle_fc_forward(nets[agent], D[agent], obs, logits, &status);
        if (a < 0) a = 0; /* the reference raises; status carries the fact */
        float top = -INFINITY, second = -INFINITY;
        for (int i = 0; i < NACT; ++i) {
            if (logits[i] > top) { second = top; top = logits[i]; }
            else if (logits[i] > second) second = logits[i];
        }
        if (top - second < min_margin) min_margin = top - second;
        if (actions_out) actions_out[timesteps] = a;
        /* env.step(action) */
        int cur = sel, nxt = (cur + 1) % 3;
        sel = nxt;
        act[cur] = a;
        if (nxt == 0) {
            double rg, ra;
            oracle_mpe_world_step(&s, act, &rg, &ra);
            rew[0] = ra; rew[1] = rg; rew[2] = rg;
            if (++world_steps >= max_cycles) trunc = 1;
        } else {
            rew[0] = rew[1] = rew[2] = 0.0;
        }
        cum[cur] = 0;
        for (int i = 0; i < 3; ++i) cum[i] += rew[i];
        /* env.last(): the NEXT agent's cumulative reward goes to the agent that just acted */
        acc[agent] += cum[sel];
        ++timesteps;
        if (limit >= 0 && timesteps >= limit) break;
        if (trunc) break;
    }
    rewards_out[0] = acc[1];
    rewards_out[1] = acc[2];
    rewards_out[2] = acc[0];
    if (min_margin_out) *min_margin_out = min_margin;
    if (status_out) *status_out = status;
    return timesteps;
}

/* ------------------------------------------------------------------ fitness sharing */
/* utils/game_logic_functions.py:12-37.  w = [n][stride] rows of flat weights; only the `cnt`
 * Linear-layer entries listed by (seg_off, seg_len) pairs take part (get_weights_ES default layers =
 * fc1, fc2, output; LayerNorm excluded).  Distances are returned too (fp32 like np.linalg.norm on
 * fp32 input; the accumulation order is this oracle's own: sequential fp32). */
double oracle_diversity(const float *individual, const float *w, int n, size_t stride,
                        const int *seg_off, const int *seg_len, int nseg, float *dist_out)
{
    float *d = (float *)malloc(sizeof(float) * n);
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int g = 0; g < nseg; ++g)
            for (int k = 0; k < seg_len[g]; ++k) {
                float df = w[(size_t)i * stride + seg_off[g] + k] - individual[seg_off[g] + k];
                s += (double)df * (double)df;
            }
        d[i] = (float)sqrt(s);
        if (dist_out) dist_out[i] = d[i];
    }
    double mean = 0.0;
    for (int i = 0; i < n; ++i) mean += d[i];
    float sigma = (float)(mean / n);
    double score = 0.0;
    for (int i = 0; i < n; ++i) {
        float sh = 1.0f - d[i] / sigma;
        if (sh > 0.0f) score += sh;
    }
    free(d);
    return score;
}

/* ------------------------------------------------------------------ counter-based Gaussian noise */
#define COEVO_NOISE_ROUNDS 7   /* offspring noise: Philox4x32-7 (csrc/philox.hip.h); the synthetic env keeps 10 rounds */
static void philox4x32(int rounds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                       uint32_t out[4])
{
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* the generator as such (known-answer tests hold it against Random123's published vectors) */
void oracle_philox4x32(int rounds, const uint32_t ctr_key[6], uint32_t out[4])
{
    philox4x32(rounds, ctr_key[0], ctr_key[1], ctr_key[2], ctr_key[3], ctr_key[4], ctr_key[5], out);
}
int oracle_noise_rounds(void) { return COEVO_NOISE_ROUNDS; }

static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4])
{
    philox4x32(10, c0, c1, c2, c3, k0, k1, out);
}

static float u32_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t float_as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ln(x) for normal x in (0,1): exponent split by bit ops + cephes logf polynomial, fmaf only. */
static float canon_logf(float x)
{
    uint32_t b = float_as_u32(x);
    int e = (int)((b >> 23) & 0xff) - 126;
    float m = u32_as_float((b & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = fmaf(y, m, -1.1514610310E-1f);
    y = fmaf(y, m, 1.1676998740E-1f);
    y = fmaf(y, m, -1.2420140846E-1f);
    y = fmaf(y, m, 1.4249322787E-1f);
    y = fmaf(y, m, -1.6668057665E-1f);
    y = fmaf(y, m, 2.0000714765E-1f);
    y = fmaf(y, m, -2.4999993993E-1f);
    y = fmaf(y, m, 3.3333331174E-1f);
    y = (y * m) * z;
    float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

/* (cos, sin) of 2*pi*u for u = k/2^24: exact quadrant split, cephes polynomials on [0, pi/4]. */
static void canon_sincos2pi(float u, float *c_out, float *s_out)
{
    float t = u * 4.0f;
    float qf = floorf(t);
    int q = (int)qf;
    float f = t - qf; /* exact, [0,1) */
    int swap = f > 0.5f;
    if (swap) f = 1.0f - f;
    float x = f * 1.57079632679489661923f;
    float z = x * x;
    float sp = -1.9515295891E-4f;
    sp = fmaf(sp, z, 8.3321608736E-3f);
    sp = fmaf(sp, z, -1.6666654611E-1f);
    float s = fmaf(sp * z, x, x);
    float cp = 2.443315711809948E-005f;
    cp = fmaf(cp, z, -1.388731625493765E-003f);
    cp = fmaf(cp, z, 4.166664568298827E-002f);
    float c = fmaf(cp * z, z, fmaf(-0.5f, z, 1.0f));
    if (swap) { float tmp = s; s = c; c = tmp; }
    switch (q & 3) {
    case 0: *c_out = c; *s_out = s; break;
    case 1: *c_out = -s; *s_out = c; break;
    case 2: *c_out = -c; *s_out = -s; break;
    default: *c_out = s; *s_out = -c; break;
    }
}

static void box_muller(uint32_t a, uint32_t b, float *z0, float *z1)
{
    float u1 = (float)(2u * (a >> 9) + 1u) * 5.9604644775390625e-08f; /* (2m+1)/2^24, exact */
    float u2 = (float)(b >> 8) * 5.9604644775390625e-08f;              /* k/2^24, exact */
    float r = sqrtf(-2.0f * canon_logf(u1));
    float c, s;
    canon_sincos2pi(u2, &c, &s);
    *z0 = r * c;
    *z1 = r * s;
}

/* Standard normals for flat indices [4*q, 4*q+4) of stream (stream_lo, stream_hi) under `seed`. */
void oracle_philox_normal4(uint64_t seed, uint32_t stream_lo, uint32_t stream_hi, uint32_t q, float z[4])
{
    uint32_t o[4];
    philox4x32(COEVO_NOISE_ROUNDS, q, stream_lo, stream_hi, 0x636f6576u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    box_muller(o[0], o[1], &z[0], &z[1]);
    box_muller(o[2], o[3], &z[2], &z[3]);
}

/* child[p] = parent[p] + sigma*z[p]  (noise rounded first, then added - agent.py:28-29).
 * skip = list of [off,len) segments left untouched (LayerNorm affine for ES, agent.py:51-53 via
 * MPE/fcnetwork.py:185-199). */
/* negate != 0: the antithetic partner, child = parent - sigma*eps (cfg 3 extension mode, not in the reference) */
void oracle_perturb_philox(const float *parent, float *child, int P, float sigma, uint64_t seed,
                           uint32_t stream_lo, uint32_t stream_hi, const int *skip_off,
                           const int *skip_len, int nskip, int negate)
{
    for (int q = 0; q * 4 < P; ++q) {
        float z[4];
        oracle_philox_normal4(seed, stream_lo, stream_hi, (uint32_t)q, z);
        for (int i = 0; i < 4 && q * 4 + i < P; ++i) {
            int p = q * 4 + i, skipped = 0;
            for (int g = 0; g < nskip; ++g)
                if (p >= skip_off[g] && p < skip_off[g] + skip_len[g]) skipped = 1;
            const float noise = sigma * z[i];
            child[p] = skipped ? parent[p] : parent[p] + (negate ? -noise : noise);
        }
    }
}

/* theta[p] += scale * sum_i fitness[i] * (pert_i[p] - theta[p]), i ascending, one fmaf per term, fp32
 * (evolutionary_strategy.py:144 with the perturbation read back from the perturbed nets).  pert = [n][P] flat. */
/* chunks: the build's canonical ES summation (include/coevo.h, coevo_es_partial): chunk c = [c*n/C, (c+1)*n/C) summed
 * i-ascending from 0, chunk sums added left to right; chunks = 1 is the plain sequential sum */
void oracle_es_update_from_pert(float *theta, int P, const float *pert, const float *fitness, int n, float scale,
                                const int *skip_off, const int *skip_len, int nskip, int chunks)
{
    for (int p = 0; p < P; ++p) {
        int skipped = 0;
        for (int g = 0; g < nskip; ++g)
            if (p >= skip_off[g] && p < skip_off[g] + skip_len[g]) skipped = 1;
        if (skipped) continue;
        float tot = 0.0f;
        for (int c = 0; c < chunks; ++c) {
            const int lo = (int)((long long)c * n / chunks), hi = (int)((long long)(c + 1) * n / chunks);
            float acc = 0.0f;
            for (int i = lo; i < hi; ++i) acc = fmaf(fitness[i], pert[(size_t)i * P + p] - theta[p], acc);
            tot = (c == 0) ? acc : tot + acc;
        }
        theta[p] = theta[p] + scale * tot;
    }
}

/* ------------------------------------------------------------------ DeepQN.forward (Atari/deepqn.py:39-48)
 * flat parameter order = torch parameters() order of DeepQN (Atari/deepqn.py:16-37): conv1.w[32][C][8][8] conv1.b
 * conv2.w[64][32][4][4] conv2.b conv3.w[64][64][3][3] conv3.b fc1.w[512][3136] fc1.b output.w[n][512] output.b
 * vbn1.w vbn1.b vbn2.w vbn2.b vbn3.w vbn3.b.
 * frame: uint8 [84][84][C] (HWC, what the env hands over; preprocess_observation's permute is an index change).
 * Canonical order: x = u8 / 255.0f; conv = bias, then fmaf over taps in (ci, ky, kx) order; BatchNorm in TRAINING
 * mode with batch 1 = per-sample, per-channel statistics over the spatial positions: S = 64 strided sums
 * (lane l adds positions l, l + 64, ... left to right, zero-padded) combined by the canonical 64-wide tree (the order the
 * GPU's packed butterfly has - the reference's own order is torch's, pinned to tolerance by the logits fixture);
 * mean = S / N, var = S2 / N
 * (biased), rstd = 1/sqrtf(var + 1e-5f), y = fmaf(d * rstd, gamma, beta), ReLU; fc = sequential-k chains. */
static float reduce_strided64(const float *v, int n)
{
    float lane[64];
    for (int l = 0; l < 64; ++l) {
        float s = (l < n) ? v[l] : 0.0f;
        for (int b = 1; b * 64 < n; ++b) s = s + ((b * 64 + l < n) ? v[b * 64 + l] : 0.0f);
        lane[l] = s;
    }
    return block_tree64(lane, 64);
}

static void conv_bn_relu(const float *in, int cin, int hin, const float *w, const float *b, const float *gamma,
                         const float *beta, int cout, int k, int stride, int hout, float *out)
{
    const int npos = hout * hout;
    float *tmp = (float *)malloc(sizeof(float) * npos);
    for (int co = 0; co < cout; ++co) {
        float *o = out + (size_t)co * npos;
        for (int oy = 0; oy < hout; ++oy)
            for (int ox = 0; ox < hout; ++ox) {
                float acc = b[co];
                for (int ci = 0; ci < cin; ++ci)
                    for (int ky = 0; ky < k; ++ky)
                        for (int kx = 0; kx < k; ++kx)
                            acc = fmaf(w[(((size_t)co * cin + ci) * k + ky) * k + kx],
                                       in[((size_t)ci * hin + oy * stride + ky) * hin + ox * stride + kx], acc);
                o[oy * hout + ox] = acc;
            }
        float mean = reduce_strided64(o, npos) / (float)npos;
        for (int p = 0; p < npos; ++p) { o[p] = o[p] - mean; tmp[p] = o[p] * o[p]; }
        float var = reduce_strided64(tmp, npos) / (float)npos;
        float rstd = 1.0f / sqrtf(var + LN_EPS);
        for (int p = 0; p < npos; ++p) {
            float y = fmaf(o[p] * rstd, gamma[co], beta[co]);
            o[p] = (y > 0.0f) ? y : (isnan(y) ? y : 0.0f);
        }
    }
    free(tmp);
}

long oracle_dqn_param_count(int C, int n)
{
    return 32L * C * 64 + 32 + 64L * 32 * 16 + 64 + 64L * 64 * 9 + 64 + 512L * 3136 + 512 + 512L * n + n + 2 * (32 + 64 + 64);
}

int oracle_dqn_forward(const float *p, int C, int n, const unsigned char *frame, float *logits)
{
    const float *w1 = p, *b1 = w1 + 32 * C * 64, *w2 = b1 + 32, *b2 = w2 + 64 * 32 * 16, *w3 = b2 + 64,
                *b3 = w3 + 64 * 64 * 9, *wf = b3 + 64, *bf = wf + 512 * 3136, *wo = bf + 512, *bo = wo + 512 * n,
                *g1 = bo + n, *be1 = g1 + 32, *g2 = be1 + 32, *be2 = g2 + 64, *g3 = be2 + 64, *be3 = g3 + 64;
    float *x = (float *)malloc(sizeof(float) * C * 84 * 84);
    float *a1 = (float *)malloc(sizeof(float) * 32 * 400), *a2 = (float *)malloc(sizeof(float) * 64 * 81);
    float *a3 = (float *)malloc(sizeof(float) * 3136), *h = (float *)malloc(sizeof(float) * 512);
    for (int y = 0; y < 84; ++y)
        for (int xx = 0; xx < 84; ++xx)
            for (int c = 0; c < C; ++c) x[((size_t)c * 84 + y) * 84 + xx] = (float)frame[((size_t)y * 84 + xx) * C + c] / 255.0f;
    conv_bn_relu(x, C, 84, w1, b1, g1, be1, 32, 8, 4, 20, a1);
    conv_bn_relu(a1, 32, 20, w2, b2, g2, be2, 64, 4, 2, 9, a2);
    conv_bn_relu(a2, 64, 9, w3, b3, g3, be3, 64, 3, 1, 7, a3);
    linear_seq(wf, bf, a3, h, 512, 3136);
    for (int j = 0; j < 512; ++j) h[j] = (h[j] > 0.0f) ? h[j] : (isnan(h[j]) ? h[j] : 0.0f);
    linear_seq(wo, bo, h, logits, n, 512);
    int best = -1;
    float cur = -INFINITY;
    for (int i = 0; i < n; ++i)
        if (logits[i] > cur) { cur = logits[i]; best = i; }
    free(x); free(a1); free(a2); free(a3); free(h);
    return best;
}

/* ------------------------------------------------------------------ synthetic Atari-shaped env + play_atari ------
 * The build's stand-in for pettingzoo.atari (no ALE in the image): see include/coevo.h, coevo_synth_step.  Stated here
 * literally - an AEC env with PettingZoo's _cumulative_rewards bookkeeping driven by the loop of play_atari
 * (utils/game_logic_functions.py:84-119) - so that the kernel's closed form credited(t) = hit(t-1) - hit(t) is checked
 * against the statement it was derived from. */
uint32_t oracle_synth_target(uint64_t seed, int64_t ordinal, int t, int n_actions)
{
    uint32_t o[4];
    philox4x32_10(0xFFFFFFFFu, (uint32_t)t, (uint32_t)ordinal, (uint32_t)((uint64_t)ordinal >> 32) ^ 0x74617267u,
                  (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return o[0] % (uint32_t)n_actions;
}

void oracle_synth_frame(uint64_t seed, int64_t ordinal, int t, int last_action, int C, unsigned char *frame)
{
    const int nbytes = 84 * 84 * C;
    for (int i = 0; i < nbytes / 16; ++i) {
        uint32_t o[4];
        philox4x32_10((uint32_t)i, (uint32_t)t | ((uint32_t)last_action << 16), (uint32_t)ordinal,
                      (uint32_t)((uint64_t)ordinal >> 32) ^ 0x66726d65u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
        memcpy(frame + 16 * (size_t)i, o, 16);   /* little endian words, as the device stores them */
    }
}

/* play_atari: returns the number of agent-steps; rewards[0] = first_0, rewards[1] = second_0 */
int oracle_dqn_play_game(const float *net_first, const float *net_second, int C, int n_actions, uint64_t seed,
                         int64_t ordinal, int limit, double rewards[2], int *actions_out)
{
    unsigned char *frame = (unsigned char *)malloc((size_t)84 * 84 * C);
    float logits[64];
    double cum[2] = {0.0, 0.0};           /* env._cumulative_rewards */
    int last = 0xFF, timesteps = 0;
    rewards[0] = rewards[1] = 0.0;
    for (int t = 0;; ++t) {                /* for agent in env.agent_iter(): the synthetic env never terminates */
        const int agent = t & 1, other = agent ^ 1;
        oracle_synth_frame(seed, ordinal, t, last, C, frame);                     /* env.observe(agent) */
        const int action = oracle_dqn_forward(agent ? net_second : net_first, C, n_actions, frame, logits);
        if (actions_out) actions_out[t] = action;
        /* env.step(action) */
        cum[agent] = 0.0;
        const double hit = ((uint32_t)action == oracle_synth_target(seed, ordinal, t, n_actions)) ? 1.0 : 0.0;
        cum[agent] += hit;
        cum[other] += -hit;
        last = action;
        /* _, reward, ... = env.last(): agent_selection has advanced to `other` */
        rewards[agent] += cum[other];
        timesteps += 1;
        if (limit >= 0 && timesteps >= limit) break;
    }
    free(frame);
    return timesteps;
}

int oracle_version(void) { return 1; }
