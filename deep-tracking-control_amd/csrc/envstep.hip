// Env-step kernels either side of the planner / the rollout store, gfx950.
//
//   dtc_compute_observations  legged_gym/envs/base/legged_robot_dtc.py:255-288  (obs_buf, privileged_obs_buf, heights)
//   dtc_check_termination     legged_gym/envs/base/legged_robot_dtc.py:229-248
//   dtc_store_transition      rsl_rl/rsl_rl/storage/rollout_storage.py:99-116 (13 copy_) + ppo.py:162-163 (time-out
//                             bootstrap) as ONE launch
//   dtc_history_roll          rsl_rl/rsl_rl/env/wrappers/history_wrapper.py:23 (cat(hist[:, D:], obs))
//
// All HBM-bound, one 64-lane wavefront per env row, 4 rows per workgroup.  Compiled with -ffp-contract=off: every
// value is produced by the same single-rounded fp32 operations, in the same order, as the torch expressions
// (oracle/observations.py), so outputs match bit for bit; the 273-point base-height mean uses the lane-strided +
// xor-butterfly order the oracle fixes.
#include "common.hpp"
#include "wave.hpp"

namespace {

struct ObsArgs {
    const float *ang_vel, *gravity, *commands, *dof_pos, *default_dof_pos, *dof_vel, *actions, *foothold_obs;
    const float *root_states, *heights_in, *forces, *noise_offset, *u_obs, *noise_scale, *u_heights;
    long long ld_forces;
    float* obs;
    float* priv;
    float* heights;
    DtcObsCfg cfg;
    int N;
    const unsigned char* where;      // NULL: all rows; else only rows with where[n] != 0
};

__global__ __launch_bounds__(256) void env_observations_kernel(const ObsArgs a) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= a.N) return;
    if (a.where && !a.where[n]) return;
    const DtcObsCfg& c = a.cfg;
    const int D = c.num_dof, F = c.num_foothold_obs, P = c.num_points;
    const int n_obs = 9 + 3 * D + F;
    // ---- proprioceptive observation: one lane per element
    if (lane < n_obs || lane + 64 < n_obs) {
        for (int e = lane; e < n_obs; e += 64) {
            float v;
            if (e < 3) v = a.ang_vel[n * 3 + e] * c.ang_vel;
            else if (e < 6) v = a.gravity[n * 3 + (e - 3)];
            else if (e < 9) v = a.commands[n * 4 + (e - 6)] * c.commands_scale[e - 6];
            else if (e < 9 + D) v = (a.dof_pos[n * D + (e - 9)] - a.default_dof_pos[e - 9]) * c.dof_pos;
            else if (e < 9 + 2 * D) v = a.dof_vel[n * D + (e - 9 - D)] * c.dof_vel;
            else if (e < 9 + 3 * D) v = a.actions[n * D + (e - 9 - 2 * D)];
            else v = a.foothold_obs[n * F + (e - 9 - 3 * D)];
            if (a.u_obs) v = v + (2.0f * a.u_obs[(long long)n * n_obs + e] - 1.0f) * a.noise_scale[e];
            a.obs[(long long)n * n_obs + e] = v;
        }
    }
    // ---- height observations: privileged = [noisy heights | force | clean heights]
    const float zt = a.root_states[n * 13 + 2] - c.base_height_target;
    const float* mh = a.heights_in + (long long)n * P;
    float* pv = a.priv + (long long)n * (2 * P + 3);
    for (int p = lane; p < P; p += 64) {
        const float h = fminf(fmaxf(zt - mh[p], -1.0f), 1.0f) * c.height_measurements;
        float nz = h;
        if (a.u_heights) nz = nz + (2.0f * a.u_heights[(long long)n * P + p] - 1.0f) * c.height_noise;
        if (a.noise_offset) nz = nz + a.noise_offset[(long long)n * P + p];
        pv[p] = nz;
        pv[P + 3 + p] = h;
        if (a.heights) a.heights[(long long)n * P + p] = h;
    }
    if (lane < 3) pv[P + lane] = a.forces[(long long)n * a.ld_forces + lane] * c.force;
}

__global__ __launch_bounds__(256) void check_termination_kernel(const float* __restrict__ contact_forces, int num_bodies,
                                                                const int* __restrict__ term_idx, int n_term,
                                                                const long long* __restrict__ episode_length,
                                                                long long max_episode_length,
                                                                const float* __restrict__ gravity,
                                                                const float* __restrict__ root_states,
                                                                const float* __restrict__ heights, DtcObsCfg c,
                                                                unsigned char* __restrict__ reset_buf,
                                                                unsigned char* __restrict__ time_out_buf,
                                                                float* __restrict__ mean_out, int N) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    bool hit = false;
    for (int j = lane; j < n_term; j += 64) {
        const float* f = contact_forces + ((long long)n * num_bodies + term_idx[j]) * 3;
        hit |= sqrtf((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]) > 100.0f;
    }
    const bool contact = __ballot(hit) != 0ull;
    const float z = root_states[n * 13 + 2];
    const float* mh = heights + (long long)n * c.num_points;
    float part = 0.f;
    bool first = true;
    for (int p = c.term_row0 + lane; p < c.term_row1; p += 64) {
        const float d = z - fmaxf(mh[p], -0.0f);
        part = first ? d : part + d;
        first = false;
    }
    const float mean = wave_sum(part) / (float)(c.term_row1 - c.term_row0);
    if (lane == 0) {
        const bool to = episode_length[n] > max_episode_length;
        const bool r = contact || to || gravity[n * 3 + 2] > 0.2f || mean < c.term_height;
        reset_buf[n] = r ? 1 : 0;
        if (time_out_buf) time_out_buf[n] = to ? 1 : 0;
        if (mean_out) mean_out[n] = mean;
    }
}

struct CopyArgs {
    DtcRowCopy item[16];
    int n_items, N;
    const float *rewards, *values;
    const unsigned char* time_outs;
    float* rewards_dst;
    float gamma;
};

// grid = (row chunks, items + 1): item y copies its rows bytewise (dword path when everything is 4-byte aligned);
// the extra y slot writes rewards + gamma * values * time_out (ppo.py:162-163)
__global__ __launch_bounds__(256) void store_transition_kernel(const CopyArgs a) {
    const int it = blockIdx.y;
    if (it == a.n_items) {
        const int n = blockIdx.x * 256 + threadIdx.x;
        if (a.rewards_dst && n < a.N) {
            float r = a.rewards[n];
            if (a.time_outs) r = r + (a.gamma * a.values[n]) * (a.time_outs[n] ? 1.0f : 0.0f);
            a.rewards_dst[n] = r;
        }
        return;
    }
    const DtcRowCopy c = a.item[it];
    const unsigned char* src = (const unsigned char*)c.src;
    unsigned char* dst = (unsigned char*)c.dst;
    const long long total = (long long)a.N * c.width_bytes;
    const bool words = ((c.width_bytes | c.src_stride_bytes) & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0;
    if (words) {
        const int w = c.width_bytes >> 2;
        const long long tw = total >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < tw; i += (long long)gridDim.x * 256) {
            const long long r = i / w;
            const int col = (int)(i - r * w);
            ((unsigned int*)dst)[i] = *(const unsigned int*)(src + r * c.src_stride_bytes + 4ll * col);
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long r = i / c.width_bytes;
            const int col = (int)(i - r * c.width_bytes);
            dst[i] = src[r * c.src_stride_bytes + col];
        }
    }
}

// roll of a [N, L*D] history: one wave reads its whole row into registers (<= 16 floats per lane), then writes it
// shifted by D with the new observation appended; `out` may alias `hist` (in place) because of that order
__global__ __launch_bounds__(256) void history_roll_kernel(const float* hist, const float* __restrict__ obs, float* out,
                                                           const unsigned char* __restrict__ reset, int N, int L, int D) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int W = L * D;
    const float* row = hist + (long long)n * W;
    float* dst = out + (long long)n * W;
    const bool zero = reset && reset[n];
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int p = lane + 64 * i + D;          // source element of destination p - D
        v[i] = 0.f;
        if (p < W) v[i] = zero ? 0.f : row[p];
        else if (p - W < D && p - D < W) v[i] = obs[(long long)n * D + (p - W)];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = lane + 64 * i;
        if (q < W) dst[q] = v[i];
    }
}

}  // namespace

extern "C" int dtc_compute_observations(const float* base_ang_vel, const float* projected_gravity, const float* commands,
                                        const float* dof_pos, const float* default_dof_pos, const float* dof_vel,
                                        const float* actions, const float* foothold_obs, const float* root_states,
                                        const float* measured_heights, const float* forces, int64_t ld_forces,
                                        const float* height_noise_offset, const float* u_obs, const float* noise_scale_vec,
                                        const float* u_heights, const DtcObsCfg* cfg, float* obs_buf,
                                        float* privileged_obs_buf, float* heights, int N, void* stream) {
    return dtc_compute_observations_where(base_ang_vel, projected_gravity, commands, dof_pos, default_dof_pos, dof_vel, actions,
                                          foothold_obs, root_states, measured_heights, forces, ld_forces, height_noise_offset, u_obs,
                                          noise_scale_vec, u_heights, cfg, obs_buf, privileged_obs_buf, heights, nullptr, N, stream);
}

extern "C" int dtc_compute_observations_where(const float* base_ang_vel, const float* projected_gravity, const float* commands,
                                              const float* dof_pos, const float* default_dof_pos, const float* dof_vel,
                                              const float* actions, const float* foothold_obs, const float* root_states,
                                              const float* measured_heights, const float* forces, int64_t ld_forces,
                                              const float* height_noise_offset, const float* u_obs, const float* noise_scale_vec,
                                              const float* u_heights, const DtcObsCfg* cfg, float* obs_buf,
                                              float* privileged_obs_buf, float* heights, const uint8_t* where, int N, void* stream) {
    DTC_REQUIRE(N >= 0 && cfg, "bad arguments");
    if (N == 0) return DTC_OK;
    DTC_REQUIRE(base_ang_vel && projected_gravity && commands && dof_pos && default_dof_pos && dof_vel && actions &&
                foothold_obs && root_states && measured_heights && forces && obs_buf && privileged_obs_buf, "null pointer");
    DTC_REQUIRE(!u_obs || noise_scale_vec, "observation noise needs noise_scale_vec");
    DTC_REQUIRE(cfg->num_dof > 0 && cfg->num_points > 0 && cfg->num_foothold_obs >= 0 && ld_forces >= 3, "bad config");
    ObsArgs a{base_ang_vel, projected_gravity, commands, dof_pos, default_dof_pos, dof_vel, actions, foothold_obs,
              root_states, measured_heights, forces, height_noise_offset, u_obs, noise_scale_vec, u_heights,
              (long long)ld_forces, obs_buf, privileged_obs_buf, heights, *cfg, N, where};
    hipStream_t s = (hipStream_t)stream;
    const double bytes = (double)N * (cfg->num_points * 4.0 * 6 + 400.0);
    dtc::ProfScope prof("compute_observations", bytes, s);
    hipLaunchKernelGGL(env_observations_kernel, dim3((unsigned)dtc::ceil_div(N, 4)), dim3(256), 0, s, a);
    return dtc::check_launch("compute_observations");
}

extern "C" int dtc_check_termination(const float* contact_forces, int num_bodies, const int32_t* termination_contact_indices,
                                     int n_term, const int64_t* episode_length_buf, int64_t max_episode_length,
                                     const float* projected_gravity, const float* root_states,
                                     const float* measured_heights, const DtcObsCfg* cfg, uint8_t* reset_buf,
                                     uint8_t* time_out_buf, float* height_mean_or_null, int N, void* stream) {
    DTC_REQUIRE(N >= 0 && cfg, "bad arguments");
    if (N == 0) return DTC_OK;
    DTC_REQUIRE(contact_forces && episode_length_buf && projected_gravity && root_states && measured_heights && reset_buf,
                "null pointer");
    DTC_REQUIRE(n_term >= 0 && (n_term == 0 || termination_contact_indices), "bad termination index list");
    DTC_REQUIRE(cfg->term_row0 >= 0 && cfg->term_row1 > cfg->term_row0 && cfg->term_row1 <= cfg->num_points, "bad height slice");
    hipStream_t s = (hipStream_t)stream;
    dtc::ProfScope prof("check_termination", (double)N * ((cfg->term_row1 - cfg->term_row0) * 4.0 + n_term * 12.0 + 40.0), s);
    hipLaunchKernelGGL(check_termination_kernel, dim3((unsigned)dtc::ceil_div(N, 4)), dim3(256), 0, s, contact_forces,
                       num_bodies, termination_contact_indices, n_term, (const long long*)episode_length_buf,
                       (long long)max_episode_length, projected_gravity, root_states, measured_heights, *cfg, reset_buf,
                       time_out_buf, height_mean_or_null, N);
    return dtc::check_launch("check_termination");
}

extern "C" int dtc_store_transition(const DtcRowCopy* items, int n_items, const float* rewards, const float* values,
                                    const uint8_t* time_outs_or_null, float gamma, float* rewards_dst, int N, void* stream) {
    DTC_REQUIRE(N >= 0 && n_items >= 0 && n_items <= 16, "bad arguments (at most 16 row copies)");
    if (N == 0) return DTC_OK;
    DTC_REQUIRE(n_items == 0 || items, "null item list");
    DTC_REQUIRE(!rewards_dst || (rewards && (!time_outs_or_null || values)), "reward write needs rewards (and values)");
    CopyArgs a{};
    double bytes = 0;
    for (int i = 0; i < n_items; ++i) {
        DTC_REQUIRE(items[i].src && items[i].dst && items[i].width_bytes > 0 && items[i].src_stride_bytes >= 0,
                    "row copy %d: bad descriptor", i);
        a.item[i] = items[i];
        bytes += 2.0 * N * items[i].width_bytes;
    }
    a.n_items = n_items;
    a.N = N;
    a.rewards = rewards;
    a.values = values;
    a.time_outs = time_outs_or_null;
    a.rewards_dst = rewards_dst;
    a.gamma = gamma;
    hipStream_t s = (hipStream_t)stream;
    dtc::ProfScope prof("store_transition", bytes, s);
    const unsigned gx = (unsigned)dtc::ceil_div(N, 256) > 64u ? (unsigned)dtc::ceil_div(N, 256) : 64u;
    hipLaunchKernelGGL(store_transition_kernel, dim3(gx, (unsigned)n_items + 1), dim3(256), 0, s, a);
    return dtc::check_launch("store_transition");
}

extern "C" int dtc_history_roll(const float* obs_history, const float* obs, float* out, const uint8_t* reset_or_null, int N,
                                int history_len, int num_obs, void* stream) {
    DTC_REQUIRE(N >= 0 && history_len >= 1 && num_obs >= 1, "bad shape");
    DTC_REQUIRE(history_len * num_obs <= 1024, "history row of %d floats exceeds 1024", history_len * num_obs);
    if (N == 0) return DTC_OK;
    DTC_REQUIRE(obs_history && obs && out, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    dtc::ProfScope prof("history_roll", (double)N * (2.0 * history_len + 1.0) * num_obs * 4.0, s);
    hipLaunchKernelGGL(history_roll_kernel, dim3((unsigned)dtc::ceil_div(N, 4)), dim3(256), 0, s, obs_history, obs, out,
                       reset_or_null, N, history_len, num_obs);
    return dtc::check_launch("history_roll");
}
