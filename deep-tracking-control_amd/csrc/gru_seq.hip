// GRU recurrence as ONE persistent launch over all T time steps (round 6): the "LDS-staged sequence tiles" of BASELINE configs[2] for
// torch.nn.GRU's recurrence of rsl_rl/rsl_rl/modules/actor_critic_recurrent.py:92-116 under ppo.py:265-335.  OPT-IN (DTC_GRU_SEQ=1 /
// dtc_set_gru_seq(1)): measured no faster in the trainers than the per-step launches on two lanes -- DESIGN.md 4.3d has the numbers.
//
// Why.  A time step of the recurrence is a [R ~ 1500] x [3H = 1536] x [H = 512] product + gate math: ~8 us of matrix work that takes ~30 us
// as a launch of its own (launch + dependency gaps, a K loop that waits for L2 on every stage, 150-300 workgroups on 256 CUs); 2 x 24 x 20
// such steps per update are 74 of the 92 ms of the recurrent workloads.
//
// How.  The recurrence couples the hidden UNITS of one row (trajectory), not the rows.  A workgroup owns (row block rb, unit tile ut): RB
// rows (one wave per 32 rows, at most 8 waves) x 16 hidden units = 48 gate columns (r | z | n of those units).  Its slice of W_hh -- 48
// rows x H, as two-term fp16 (hi, lo) scaled by one power of two, 99 KiB -- is built ONCE in LDS and serves all T steps.  Per step a wave
// reads its 32 rows of h_{t-1} (two-term fp16, fixed scale 2^14: |h| <= 1) straight from the exchange buffer into MFMA operand registers
// (16-byte loads running 4 K steps ahead; no LDS staging: rows are not shared between waves), runs 3 v_mfma_f32_16x16x32_f16 passes per
// product (lo hi', hi lo', hi hi': exact in the fp32 accumulator) with W as the A operand, so that a lane ends up with FOUR CONSECUTIVE
// UNITS of one row: float4 loads of gi, float4 stores of h_t / gates / gh_n, 8-byte stores of the next step's operand.  The 32 workgroups
// of a row block then meet at a counter barrier (agent-scope atomics): the exchange rows leave by write-through (agent-scope) stores BEFORE
// the arrival, the step's fp32 outputs and the next step's gi ride under the wait, and one wave per workgroup invalidates the CU's L1 (and
// stale L2 lines) after the barrier -- the 16-32 workgroups of a row block that share an XCD then share its L2 for the re-reads.
// Double-buffered exchange, one barrier per step, no grid-wide synchronisation, no kernel boundary.
//
// Residency.  The barrier needs every workgroup of a row block resident at once; a workgroup takes a CU (LDS).  dtc_gru_fwd (one
// recurrence; the other one may run beside it on another stream) keeps to 4 row blocks x (H / 16) = 128 workgroups = half the CUs (R <= 1024);
// dtc_gru_fwd_multi runs both recurrences of an actor-critic one after the other inside ONE launch of up to 8 x 32 = 256 workgroups
// (R <= 2048).  Should the workgroups of a launch ever fail to meet (more such launches than the device has CUs for: several trainer
// processes on ONE device), the spin gives up after 2 s, raises the flag of dtc_gru_seq_status() and the launch ends -- a loud failure
// instead of a hang.
#include <stdlib.h>

#include "common.hpp"

namespace {

typedef _Float16 h16;
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int UT = 16;                  // hidden units per workgroup
constexpr int GC = 3 * UT;              // gate columns per workgroup
constexpr int EH = 14;                  // h is stored times 2^14 (|h| <= 1: hi < 2^15)
constexpr int MAX_RB_ROWS = 256;        // exclusive launches: 8 waves per workgroup (256 VGPRs each: the row operand's loads run 4 K steps ahead)
constexpr int MAX_RB_ROWS_HALF = 384;   // half-chip launches: 12 waves per workgroup (170 VGPRs each: loads 2 K steps ahead)
constexpr int MAX_NRB = 8;              // 8 x (H / 16) = 256 workgroups at most: one per CU

struct SeqFwd {
    const float* gi;      // [T, R, 3H]
    const float* h0;      // [R, H]
    const float* Whh;     // [3H, H]
    const float* bhh;     // [3H]
    float* hs_all;        // [T + 1, R, H]
    float* gates;         // [T, R, 3H]
    float* hn;            // [T, R, H]
    h16* hx;              // exchange: [2 buffers][2 planes][rows_pad][H]
    unsigned* bar;        // [MAX_NRB] arrival counters
    unsigned* err;        // global error flag (dtc_gru_seq_status)
    unsigned long long* trace;   // debugging (dtc_gru_seq_trace): [workgroup][T][4] time stamps (100 MHz) of thread 0 -- met / K loop / gates / arrived
    int T, R, RB, NRB, rows_pad;
};

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// (hi, lo) of four values times 2^e, as the two 8-byte words of the planes
__device__ __forceinline__ void split4(const f32x4 v, int e, u64& hi, u64& lo) {
    h16x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xs = __builtin_ldexpf(v[i], e);
        h[i] = (h16)xs;
        l[i] = (h16)(xs - (float)h[i]);
    }
    hi = __builtin_bit_cast(u64, h);
    lo = __builtin_bit_cast(u64, l);
}

__device__ __forceinline__ f32x4 load4(const float* p, bool aligned) {
    if (aligned) return *reinterpret_cast<const f32x4*>(p);
    return f32x4{p[0], p[1], p[2], p[3]};
}

// All workgroups of the row block have arrived `target` times in total.  Thread 0 spins; false = gave up (error flag raised).
__device__ __forceinline__ bool group_wait(unsigned* bar, unsigned target, unsigned* err, int* lds_flag) {
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {          // 2 s: the row block's workgroups never met
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *lds_flag = ok;
    }
    // what the other workgroups wrote before they arrived becomes visible to this CU: ONE wave drops the CU's L1 lines (and the L2 lines
    // that another XCD may have superseded); the row operand is then read by plain 16-byte loads that the 16 workgroups of a row block on
    // this XCD share in its L2 (agent-scope loads of every wave went to the fabric each time: 32 x the traffic, 58 us per step)
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    return *lds_flag != 0;
}
// Every store of this workgroup so far is visible at agent scope, then ONE arrival
__device__ __forceinline__ void group_arrive(unsigned* bar) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct SeqFwdPair {
    int count;
    SeqFwd item[2];
};

// `count` recurrences of one shape, one after the other (the actor's and the critic's memory: dtc_gru_fwd_multi)
template <int H, int MAXT, int PFC>
__global__ __launch_bounds__(MAXT) void gru_seq_fwd_kernel(const SeqFwdPair P) {
    constexpr int WROW = 2 * H + 32;                  // bytes of one LDS row of a plane (+32: the 16 lanes of a read group hit 64 different banks)
    constexpr int KS = H / 32;                        // K steps of a time step
    constexpr int PF = PFC;                           // register sets of the h operand: its loads run PF - 1 K steps ahead of the MFMAs
    __shared__ __attribute__((aligned(16))) unsigned char Wl[2][GC][WROW];
    __shared__ float red[16];
    __shared__ __attribute__((aligned(16))) float bl[3][UT];
    __shared__ int flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int c16 = lane & 15, kg = lane >> 4;
    constexpr int nut = H / UT;
  for (int q = 0; q < P.count; ++q) {
    const SeqFwd& a = P.item[q];
    const int rb = blockIdx.x % a.NRB, ut = blockIdx.x / a.NRB;
    const int R = a.R, T = a.T;
    unsigned* bar = a.bar + rb;
    __syncthreads();                                  // (the previous recurrence's last reads of Wl / bl)

    // ---- prologue 1: this workgroup's slice of W_hh -> LDS as (hi, lo) x 2^ew  (W_hh is a view into the parameter arena: any 4-byte alignment)
    const bool w16 = (reinterpret_cast<uintptr_t>(a.Whh) & 15u) == 0;
    float am = 0.f;
    for (int e = tid; e < GC * (H / 4); e += nthr) {
        const int c = e / (H / 4), k4 = e - c * (H / 4);
        const f32x4 v = load4(a.Whh + ((size_t)(c >> 4) * H + ut * UT + (c & 15)) * H + 4 * k4, w16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = fabsf(v[i]);
            am = (x < __builtin_inff() && x > am) ? x : am;          // finite values choose the scale (NaN / inf pass through as fp16 NaN / inf)
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) am = fmaxf(am, __shfl_xor(am, off, 64));
    if (lane == 0) red[wave] = am;
    __syncthreads();
    am = red[0];
    for (int w = 1; w < (nthr >> 6); ++w) am = fmaxf(am, red[w]);
    const unsigned ab = __float_as_uint(am);
    int ew = ab == 0u ? 0 : 141 - (int)(ab >> 23);                   // amax x 2^ew in [2^14, 2^15)
    ew = ew > 100 ? 100 : ew;
    for (int e = tid; e < GC * (H / 4); e += nthr) {
        const int c = e / (H / 4), k4 = e - c * (H / 4);
        const f32x4 v = load4(a.Whh + ((size_t)(c >> 4) * H + ut * UT + (c & 15)) * H + 4 * k4, w16);
        u64 hi, lo;
        split4(v, ew, hi, lo);
        *reinterpret_cast<u64*>(&Wl[0][c][8 * k4]) = hi;
        *reinterpret_cast<u64*>(&Wl[1][c][8 * k4]) = lo;
    }
    if (tid < GC) bl[tid >> 4][tid & 15] = a.bhh[(tid >> 4) * H + ut * UT + (tid & 15)];
    const float back = __builtin_ldexpf(1.0f, -(EH + ew));            // accumulator -> gh

    // ---- this lane: rows row0 + 16 rt + c16 (rt = 0, 1) as the MFMA's N index, units ut * 16 + 4 kg .. + 3 as its M index (per gate)
    const size_t plane = (size_t)a.rows_pad * H;                     // elements of one plane of one buffer
    const int row0 = rb * a.RB + 32 * wave;
    int rowc[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) rowc[rt] = row0 + 16 * rt + c16;
    const int ucol = ut * UT + 4 * kg;
    const unsigned char* wl_lane = &Wl[0][c16][16 * kg];               // + plane * GC * WROW + gate * 16 * WROW + kstep * 64

    // ---- prologue 2: h0 of this lane's (rows, units) -> hs_all[0], the exchange buffer, and registers (h_{t-1} of the gate math)
    f32x4 hp[2], gin[2][3];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int row = rowc[rt];
        hp[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < R) {
            hp[rt] = *reinterpret_cast<const f32x4*>(a.h0 + (size_t)row * H + ucol);
            *reinterpret_cast<f32x4*>(a.hs_all + (size_t)row * H + ucol) = hp[rt];
        }
        u64 hi, lo;
        split4(hp[rt], EH, hi, lo);
        __hip_atomic_store(reinterpret_cast<u64*>(a.hx + (size_t)row * H + ucol), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<u64*>(a.hx + plane + (size_t)row * H + ucol), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    group_arrive(bar);
    auto load_gi = [&](int t) {                                       // the step's input pre-activations (independent of the barrier)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = rowc[rt] < R ? rowc[rt] : R - 1;
            const float* gp = a.gi + ((size_t)t * R + row) * 3 * H + ucol;
#pragma unroll
            for (int g = 0; g < 3; ++g) gin[rt][g] = *reinterpret_cast<const f32x4*>(gp + g * H);
        }
    };
    load_gi(0);

    for (int t = 0; t < T; ++t) {
        if (!group_wait(bar, (unsigned)(nut * (t + 1)), a.err, &flag)) return;
        unsigned long long* tr = (a.trace && tid == 0) ? a.trace + ((size_t)blockIdx.x * T + t) * 4 : nullptr;
        if (tr) tr[0] = __builtin_amdgcn_s_memrealtime();
        const h16* hxr = a.hx + (size_t)(t & 1) * 2 * plane;
        h16* hxw = a.hx + (size_t)((t + 1) & 1) * 2 * plane;
        // h operand: lane = (row c16 of row tile rt, k group kg): 8 consecutive k of both planes (16 bytes each)
        h16x8 hb[PF][2][2];
        auto load_h = [&](int set, int ks) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    hb[set][rt][p] = *reinterpret_cast<const h16x8*>(hxr + (size_t)p * plane + (size_t)rowc[rt] * H + ks * 32 + 8 * kg);
        };
        f32x4 acc[2][3];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[rt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < PF - 1; ++s) load_h(s, s);
        __builtin_amdgcn_sched_barrier(0);                 // (issue order is the point: hipcc otherwise sinks the loads to their uses)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cur = ks % PF;
            if (ks + PF - 1 < KS) {
                load_h((ks + PF - 1) % PF, ks + PF - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            h16x8 hf[2][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int p = 0; p < 2; ++p) hf[rt][p] = hb[cur][rt][p];
            // gate by gate; per product smallest terms first: lo hi', hi lo', hi hi' (W is the A operand: result rows = gate columns,
            // result columns = batch rows)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                h16x8 wf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    wf[p] = *reinterpret_cast<const h16x8*>(wl_lane + (size_t)p * GC * WROW + (size_t)g * 16 * WROW + ks * 64);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], hf[rt][0], acc[rt][g], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], hf[rt][1], acc[rt][g], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], hf[rt][0], acc[rt][g], 0, 0, 0);
            }
        }
        if (tr) tr[1] = __builtin_amdgcn_s_memrealtime();
        // ---- gate math of torch.nn.GRU (gru_step_fwd_kernel): r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r gh_n),
        // h_t = (1 - z) n + z h_{t-1}; gh = h_{t-1} W_hh^T + b_hh.  The next step's operand leaves FIRST, the workgroup arrives, and only
        // then the step's fp32 outputs are stored and the next gi requested: both ride under the wait for the row block
        f32x4 bias[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) bias[g] = *reinterpret_cast<const f32x4*>(&bl[g][4 * kg]);
        f32x4 rg[2], zg[2], ng[2], ghn[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4 hnew;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rg[rt][i] = sigmoid_f(gin[rt][0][i] + (acc[rt][0][i] * back + bias[0][i]));
                zg[rt][i] = sigmoid_f(gin[rt][1][i] + (acc[rt][1][i] * back + bias[1][i]));
                ghn[rt][i] = acc[rt][2][i] * back + bias[2][i];
                ng[rt][i] = tanhf(gin[rt][2][i] + rg[rt][i] * ghn[rt][i]);
                hnew[i] = (1.0f - zg[rt][i]) * ng[rt][i] + zg[rt][i] * hp[rt][i];
            }
            if (rowc[rt] >= R) hnew = f32x4{0.f, 0.f, 0.f, 0.f};      // padding rows of the last row block stay zero
            hp[rt] = hnew;
            if (t + 1 < T) {
                u64 hi, lo;
                split4(hnew, EH, hi, lo);
                __hip_atomic_store(reinterpret_cast<u64*>(hxw + (size_t)rowc[rt] * H + ucol), hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<u64*>(hxw + plane + (size_t)rowc[rt] * H + ucol), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tr) tr[2] = __builtin_amdgcn_s_memrealtime();
        if (t + 1 < T) group_arrive(bar);
        if (tr) tr[3] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = rowc[rt];
            if (row < R) {
                float* gp = a.gates + ((size_t)t * R + row) * 3 * H + ucol;
                *reinterpret_cast<f32x4*>(gp) = rg[rt];
                *reinterpret_cast<f32x4*>(gp + H) = zg[rt];
                *reinterpret_cast<f32x4*>(gp + 2 * H) = ng[rt];
                *reinterpret_cast<f32x4*>(a.hn + ((size_t)t * R + row) * H + ucol) = ghn[rt];
                *reinterpret_cast<f32x4*>(a.hs_all + ((size_t)(t + 1) * R + row) * H + ucol) = hp[rt];
            }
        }
        if (t + 1 < T) load_gi(t + 1);
    }
  }
}

int g_seq_mode = -1;                 // -1: DTC_GRU_SEQ decides (default off), 0 / 1: dtc_set_gru_seq
unsigned long long* g_trace = nullptr;
unsigned* g_err = nullptr;           // one device word per process (never freed)

unsigned* err_word() {
    if (g_err == nullptr) {
        if (hipMalloc((void**)&g_err, sizeof(unsigned)) != hipSuccess) return nullptr;
        (void)hipMemset(g_err, 0, sizeof(unsigned));
    }
    return g_err;
}

}  // namespace

extern "C" void dtc_set_gru_seq(int on) { g_seq_mode = on < 0 ? -1 : (on != 0); }
// debugging: a device buffer of (workgroups x T x 4) 8-byte words that the NEXT launches fill with time stamps (NULL: off)
extern "C" void dtc_gru_seq_trace(void* buf) { g_trace = (unsigned long long*)buf; }
// OFF unless asked for (DTC_GRU_SEQ=1 or dtc_set_gru_seq(1)): measured in the trainers (round 6) the persistent forward is
// no faster than the two per-step chains on two lanes -- 22 us per step and recurrence against ~24 -- and merging both recurrences into one
// launch costs the overlap with the other head's MLP: 92.4 vs 87.5 ms per configs[2] step.
extern "C" int dtc_get_gru_seq(void) {
    static const bool env_on = getenv("DTC_GRU_SEQ") && atoi(getenv("DTC_GRU_SEQ")) == 1;
    return g_seq_mode < 0 ? (env_on ? 1 : 0) : g_seq_mode;
}

// 0 = every persistent launch so far ran to its end; 1 = a launch gave up at a barrier (its outputs are incomplete).  Synchronises with the
// device (4-byte copy); `reset` clears the flag.
extern "C" int dtc_gru_seq_status(int reset) {
    if (g_err == nullptr) return 0;
    unsigned v = 0;
    if (hipMemcpy(&v, g_err, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (reset && v != 0u) (void)hipMemset(g_err, 0, sizeof(unsigned));
    return (int)v;
}

// bytes the persistent kernels need behind the per-step regions of a dtc_gru_workspace() buffer
extern "C" int64_t dtc_gru_seq_workspace(int R, int H) {
    if (R <= 0 || H <= 0) return 0;
    return 8ll * ((int64_t)R + 4 * MAX_RB_ROWS_HALF) * H + 256;
}

// The geometry of a launch: (row blocks, rows per block), or false when the shape is not served.  `exclusive`: the launch may take every
// CU (8 row blocks x 32 unit tiles = 256 workgroups) -- dtc_gru_fwd_multi, ONE launch for both recurrences of an actor-critic; a single
// recurrence (dtc_gru_fwd: the other one may run beside it on another stream) keeps to 4 row blocks = 128 workgroups = half the CUs.
static bool seq_geometry(int T, int R, int H, bool exclusive, int& nrb, int& rbrows) {
    if (!dtc_get_gru_seq() || H != 512 || T < 2 || R < 1) return false;
    const int max_nrb = exclusive ? MAX_NRB : MAX_NRB / 2;
    nrb = (R + 31) / 32;
    nrb = nrb > max_nrb ? max_nrb : nrb;
    rbrows = (((R + nrb - 1) / nrb) + 31) / 32 * 32;
    return rbrows <= (exclusive ? MAX_RB_ROWS : MAX_RB_ROWS_HALF);
}
extern "C" int dtc_gru_seq_supported(int T, int R, int H, int exclusive) {
    int nrb, rbrows;
    return seq_geometry(T, R, H, exclusive != 0, nrb, rbrows) ? 1 : 0;
}

static int seq_launch(SeqFwdPair& P, int nrb, int rbrows, int T, int R, int H, hipStream_t s) {
    unsigned* err = err_word();
    DTC_REQUIRE(err != nullptr, "gru_seq_fwd: no device memory for the status word");
    for (int q = 0; q < P.count; ++q) {
        SeqFwd& a = P.item[q];
        DTC_REQUIRE(a.gi && a.h0 && a.Whh && a.bhh && a.hs_all && a.gates && a.hn && a.bar, "gru_seq_fwd: null pointer (item %d)", q);
        DTC_REQUIRE(dtc::aligned16(a.gi) && dtc::aligned16(a.h0) && dtc::aligned16(a.hs_all) && dtc::aligned16(a.gates) && dtc::aligned16(a.hn) &&
                        dtc::aligned16(a.bar), "gru_seq_fwd: gi / h0 / hs_all / gates / hn / workspace must be 16-byte aligned (item %d)", q);
        a.hx = (h16*)((char*)a.bar + 256);
        a.err = err;
        a.trace = g_trace;
        a.T = T; a.R = R; a.RB = rbrows; a.NRB = nrb; a.rows_pad = nrb * rbrows;
        if (hipMemsetAsync(a.bar, 0, 256, s) != hipSuccess) {
            dtc::set_error("gru_seq_fwd: memset failed");
            return DTC_ERR_LAUNCH;
        }
    }
    dtc::ProfScope prof("gru_seq_fwd", P.count * 2.0 * T * (double)R * 3.0 * H * H, s);
    if (rbrows <= MAX_RB_ROWS)
        hipLaunchKernelGGL((gru_seq_fwd_kernel<512, 512, 5>), dim3((unsigned)(nrb * (H / UT))), dim3((unsigned)(rbrows / 32 * 64)), 0, s, P);
    else
        hipLaunchKernelGGL((gru_seq_fwd_kernel<512, 768, 3>), dim3((unsigned)(nrb * (H / UT))), dim3((unsigned)(rbrows / 32 * 64)), 0, s, P);
    return dtc::check_launch("gru_seq_fwd");
}

// The whole forward recurrence as one launch.  `seq_ws`: dtc_gru_seq_workspace(R, H) bytes, 16-byte aligned.
extern "C" int dtc_gru_seq_fwd(const float* gi, const float* h0, const float* W_hh, const float* b_hh, float* hs_all, float* gates, float* hn,
                               void* seq_ws, int T, int R, int H, void* stream) {
    int nrb, rbrows;
    DTC_REQUIRE(seq_geometry(T, R, H, false, nrb, rbrows), "gru_seq_fwd: shape T=%d R=%d H=%d not served", T, R, H);
    SeqFwdPair P;
    P.count = 1;
    SeqFwd& a = P.item[0];
    a.gi = gi; a.h0 = h0; a.Whh = W_hh; a.bhh = b_hh; a.hs_all = hs_all; a.gates = gates; a.hn = hn;
    a.bar = (unsigned*)seq_ws;
    P.item[1] = a;
    return seq_launch(P, nrb, rbrows, T, R, H, (hipStream_t)stream);
}

// Two recurrences of one shape (the actor's and the critic's), one after the other inside ONE launch that may take every CU.
extern "C" int dtc_gru_seq_fwd_pair(const float* const* gi, const float* const* h0, const float* const* W_hh, const float* const* b_hh,
                                    float* const* hs_all, float* const* gates, float* const* hn, void* const* seq_ws, int T, int R, int H,
                                    void* stream) {
    int nrb, rbrows;
    DTC_REQUIRE(seq_geometry(T, R, H, true, nrb, rbrows), "gru_seq_fwd_pair: shape T=%d R=%d H=%d not served", T, R, H);
    DTC_REQUIRE(gi && h0 && W_hh && b_hh && hs_all && gates && hn && seq_ws, "null pointer");
    SeqFwdPair P;
    P.count = 2;
    for (int q = 0; q < 2; ++q) {
        SeqFwd& a = P.item[q];
        a.gi = gi[q]; a.h0 = h0[q]; a.Whh = W_hh[q]; a.bhh = b_hh[q]; a.hs_all = hs_all[q]; a.gates = gates[q]; a.hn = hn[q];
        a.bar = (unsigned*)seq_ws[q];
    }
    return seq_launch(P, nrb, rbrows, T, R, H, (hipStream_t)stream);
}
