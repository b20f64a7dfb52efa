/* dtc_hip.h -- C ABI of libdtc_hip.so, the MI355X (gfx950) hot path of Deep-Tracking-Control.
 *
 * The reference has no FFI: its hot path is Python calling torch.  Each entry
 * point below replaces the torch op sequence of the cited reference lines; the Python classes
 * in deep-tracking-control_amd/dtc_amd keep the reference's signatures and call these through
 * ctypes with `tensor.data_ptr()` (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is documented as host;
 *   - the caller owns all buffers; nothing is retained after return;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); calls are
 *     asynchronous on it and never synchronise the host;
 *   - return 0 on success, a negative DTC_ERR_* otherwise (bad argument / launch failure);
 *     no exception crosses the boundary;
 *   - all matrices are row-major fp32; `ld*` are leading dimensions in elements.
 */
#ifndef DTC_HIP_H
#define DTC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTC_OK 0
#define DTC_ERR_ARG (-1)      /* null pointer / bad shape / unsupported size            */
#define DTC_ERR_LAUNCH (-2)   /* hipGetLastError() != hipSuccess after a launch           */
#define DTC_ERR_ALIGN (-3)    /* a pointer that must be 16-byte aligned is not            */

#define DTC_ACT_NONE 0
#define DTC_ACT_RELU 1
#define DTC_ACT_ELU 2
/* the remaining entries of the reference's get_activation table (actor_critic_decoder.py:565-582; 'crelu' is nn.ReLU there) */
#define DTC_ACT_SELU 3
#define DTC_ACT_LRELU 4    /* nn.LeakyReLU(): negative slope 0.01 */
#define DTC_ACT_TANH 5
#define DTC_ACT_SIGMOID 6

/* library / device info ------------------------------------------------------------------ */
#define DTC_ABI_VERSION 15                   /* bumped whenever a signature or a by-value struct layout changes       */
int dtc_version(void);                       /* == DTC_ABI_VERSION of the build; the host binding refuses a mismatch   */
/* sizeof() of the structs that cross the boundary, in the order DtcGridCfg, DtcObsCfg, DtcRowCopy, DtcSeg, DtcSegMat,
   DtcFwdLayer, DtcWgradJob, DtcPpoCfg, DtcProfRec, DtcWimgJob, DtcH2iWJob, DtcH2iOperand, DtcWgradH2iJob, DtcEnvStep, DtcH2iFwdLayer, DtcH2iDgradLayer, DtcGruFwdItem, DtcGruBwdItem: the binding compares them with its own layouts at load time (a library
   built from another revision -- e.g. a stale DTC_LIB override -- must not receive descriptors it would misread).
   Returns the number of entries (written up to `cap`). */
int dtc_abi_sizes(int64_t* out, int cap);
const char* dtc_last_error(void);            /* host string describing the last failure   */
/* A non-blocking HIP stream owned by the caller (high_priority != 0: the device's greatest priority).  The trainers run their second compute
 * lane and their weight-gradient lanes on such streams instead of entries of torch's shared stream pool (dtc_amd/algorithms/ppo.py:_Lanes). */
int dtc_stream_create(int high_priority, void** out);
int dtc_stream_destroy(void* stream);

/* ---- foothold planner: legged_gym/envs/base/legged_robot_dtc.py:98-201 ----------------- */
typedef struct DtcGridCfg {
    int nx, ny;            /* 33, 21: cfg.terrain.measured_points_{x,y} (lite3_dtc_config.py:32-36) */
    float t_stance;        /* cfg.sim.dt * cfg.control.decimation = 0.02 (legged_robot_dtc.py:106)  */
    float fdbk_gain;       /* 0.03 (legged_robot_dtc.py:107)                                        */
    float x[64];           /* host copies of the grid coordinates                                   */
    float y[32];
} DtcGridCfg;

/* One call = lines :98-201 for N envs.  idx is the reference's optimal_foothold_indice [N,1,4]
 * (int64).  Optional outputs may be NULL: score [N,P,4] (self.foothold_score),
 * nominal_idx [N,4] int64, slope [N,nx,ny], heights_world [N,P,3].  cfg is a HOST pointer. */
int dtc_foothold_plan(const float* measured_heights /*[N,P]*/, const float* root_states /*[N,13]*/,
                      const float* thigh_pos /*[N,4,3]*/, const float* commands /*[N,4]*/,
                      const DtcGridCfg* cfg, int64_t* idx /*[N,4]*/, float* foothold_obs /*[N,8]*/,
                      float* opt_world /*[N,4,3]*/, float* pred /*[N,4,3]*/, float* pred_to_robot /*[N,4,3]*/,
                      float* score_or_null, int64_t* nominal_idx_or_null, float* slope_or_null,
                      float* heights_world_or_null, int N, void* stream);

/* Consumers of the planner output in the same env step (legged_robot_dtc.py:577-586 and :536-539):
 * tracking[n] = sum_legs contact ? -log(0.8 + ||foot_xy - optimal_xy||) : 0   (_reward_tracking_optimal_footholds)
 * miss[n]     = min_legs foot_z < 0 ? 1 : 0                                    (_reward_foothold_miss)
 * contact is the env's contact_filt as uint8 [N,4]; either output may be NULL. */
int dtc_foothold_rewards(const float* foot_positions /*[N,4,3]*/, const float* opt_world /*[N,4,3]*/,
                         const uint8_t* contact /*[N,4]*/, float* tracking /*[N]*/, float* miss /*[N]*/, int N,
                         void* stream);

/* LeggedRobot._get_heights: legged_gym/envs/base/legged_robot.py:1279-1317 (row f1). */
int dtc_get_heights(const int16_t* height_samples /*[rows,cols]*/, int rows, int cols,
                    const float* root_states /*[N,13]*/, const DtcGridCfg* cfg, float border_size,
                    float horizontal_scale, float vertical_scale, float* measured_heights /*[N,P]*/,
                    int N, void* stream);

/* Rows f1 + c fused: sample the terrain table (legged_robot.py:1279-1317), write measured_heights [N,P] (the observations
 * consume it) and plan the footholds (legged_robot_dtc.py:98-201) from the freshly sampled rows in ONE launch -- no
 * read-back of the [N,P] matrix.  Bit-identical to dtc_get_heights followed by dtc_foothold_plan (any grid other than
 * the reference's 33 x 21 runs exactly those two). */
int dtc_foothold_plan_from_table(const int16_t* height_samples /*[rows,cols]*/, int rows, int cols, float border_size,
                                 float horizontal_scale, float vertical_scale, const float* root_states /*[N,13]*/,
                                 const float* thigh_pos /*[N,4,3]*/, const float* commands /*[N,4]*/,
                                 const DtcGridCfg* cfg, float* measured_heights /*out [N,P]*/, int64_t* idx /*[N,4]*/,
                                 float* foothold_obs /*[N,8]*/, float* opt_world /*[N,4,3]*/, float* pred /*[N,4,3]*/,
                                 float* pred_to_robot /*[N,4,3]*/, int N, void* stream);

/* ---- env-step consumers of the planner output (row f3) ----------------------------------
 * LeggedRobotDTC.compute_observations, legged_gym/envs/base/legged_robot_dtc.py:255-288, and
 * LeggedRobotDTC.check_termination, legged_gym/envs/base/legged_robot_dtc.py:229-248. */
typedef struct DtcObsCfg {
    float ang_vel, dof_pos, dof_vel, height_measurements, force;   /* cfg.normalization.obs_scales (legged_robot_config.py:181-188) */
    float commands_scale[3];     /* [lin_vel, lin_vel, ang_vel] (legged_robot.py:810)                                    */
    float base_height_target;    /* cfg.rewards.base_height_target (lite3_dtc_config.py:139)                             */
    float height_noise;          /* 0.1: amplitude of the uniform height noise of the privileged obs (:278)              */
    float term_height;           /* 0.15: base-height termination threshold (:244-246)                                   */
    int num_dof, num_foothold_obs, num_points;      /* 12, 8, 693                                                        */
    int term_row0, term_row1;    /* slice of measured_heights averaged by the height test: 10*21, (33-10)*21             */
} DtcObsCfg;

/* obs_buf [N, 9 + 3*num_dof + num_foothold_obs] = cat(ang_vel*s, gravity, commands[:, :3]*s, (dof_pos-default)*s,
 * dof_vel*s, actions, foothold_obs) (+ (2*u_obs-1)*noise_scale_vec when u_obs != NULL: the torch.rand_like draw of :287
 * is an INPUT); heights [N,P] = clip(root_z - target - measured_heights, -1, 1)*s (may be NULL);
 * privileged_obs_buf [N, 2P+3] = cat(heights + (2*u_heights-1)*0.1 + height_noise_offset, forces[:,0,:]*s, heights)
 * (u_heights / height_noise_offset may be NULL = term omitted).  forces: first body's force of env n at
 * forces + n*ld_forces.  commands is [N,4].  cfg is a HOST pointer. */
int dtc_compute_observations(const float* base_ang_vel, const float* projected_gravity, const float* commands,
                             const float* dof_pos, const float* default_dof_pos /*[num_dof]*/, const float* dof_vel,
                             const float* actions, const float* foothold_obs, const float* root_states /*[N,13]*/,
                             const float* measured_heights /*[N,P]*/, const float* forces, int64_t ld_forces,
                             const float* height_noise_offset /*[N,P]*/, const float* u_obs, const float* noise_scale_vec,
                             const float* u_heights /*[N,P]*/, const DtcObsCfg* cfg, float* obs_buf,
                             float* privileged_obs_buf, float* heights, int N, void* stream);

/* reset_buf[n] = any_j |contact_forces[n, idx_j]| > 100  |  episode_length > max  |  gravity_z > 0.2  |
 * mean(root_z - max(measured_heights[n, term_row0:term_row1], 0)) < term_height;  time_out_buf[n] = episode_length > max.
 * contact_forces [N,num_bodies,3]; indices int32 (device); episode_length_buf int64; outputs uint8. */
int dtc_check_termination(const float* contact_forces, int num_bodies, const int32_t* termination_contact_indices,
                          int n_term, const int64_t* episode_length_buf, int64_t max_episode_length,
                          const float* projected_gravity /*[N,3]*/, const float* root_states /*[N,13]*/,
                          const float* measured_heights /*[N,P]*/, const DtcObsCfg* cfg, uint8_t* reset_buf,
                          uint8_t* time_out_buf /*or NULL*/, float* height_mean_or_null, int N, void* stream);

/* Same as dtc_compute_observations for the rows with where[n] != 0 only (the other rows of the three outputs are left untouched):
 * the observation pass AFTER `reset_idx` (legged_robot_dtc.py:209-211) when dtc_env_post_physics has already written the rows
 * of the envs that were not reset.  where == NULL: all rows. */
int dtc_compute_observations_where(const float* base_ang_vel, const float* projected_gravity, const float* commands,
                                   const float* dof_pos, const float* default_dof_pos, const float* dof_vel,
                                   const float* actions, const float* foothold_obs, const float* root_states,
                                   const float* measured_heights, const float* forces, int64_t ld_forces,
                                   const float* height_noise_offset, const float* u_obs, const float* noise_scale_vec,
                                   const float* u_heights, const DtcObsCfg* cfg, float* obs_buf, float* privileged_obs_buf,
                                   float* heights, const uint8_t* where, int N, void* stream);

/* ---- one env step's post-physics block as ONE launch (BASELINE configs[3]: 4096 envs, launch-bound) -----------------------
 * LeggedRobotDTC.post_physics_step, legged_gym/envs/base/legged_robot_dtc.py:98-211, on the env's own buffers:
 *   [measured_heights = _get_heights() (legged_robot.py:1279-1317) when height_samples != NULL]
 *   foothold block :98-201  ->  check_termination :229-248  ->  foothold rewards :577-586 / :536-539  ->  compute_observations :255-288
 * = dtc_foothold_plan[_from_table] + dtc_check_termination + dtc_foothold_rewards + dtc_compute_observations with the same
 * arguments, same bits, one launch (each env's height row is read once and stays in LDS for all four).  The reference resets
 * envs (`reset_idx`, simulator state) BETWEEN the rewards and the observations: the observation rows written here are those of
 * the state passed in; after resetting, refresh the rows of the reset envs with dtc_compute_observations_where(..., where =
 * reset_buf).  All pointers are device pointers; optional ones may be NULL where the separate entry points allow it.
 * Grids other than 33 x 21 run the separate launches in that order.  `st`, `grid`, `obs` are HOST pointers. */
typedef struct DtcEnvStep {
    /* planner */
    const int16_t* height_samples;   /* terrain table [rows, cols] or NULL (then measured_heights is an input) */
    int rows, cols;
    float border_size, horizontal_scale, vertical_scale;
    const float *root_states, *thigh_pos, *commands;
    float* measured_heights;         /* [N,P]: written when height_samples != NULL, read otherwise */
    int64_t* idx;
    float *foothold_obs, *opt_world, *pred, *pred_to_robot;
    /* check_termination */
    const float* contact_forces;
    const int32_t* termination_contact_indices;
    const int64_t* episode_length_buf;
    const float* projected_gravity;
    int64_t max_episode_length;
    int num_bodies, n_term;
    uint8_t *reset_buf, *time_out_buf;
    float* height_mean;
    /* rewards */
    const float* foot_positions;
    const uint8_t* contact_filt;
    float *rew_tracking, *rew_miss;
    /* compute_observations (commands, root_states, projected_gravity, foothold_obs, measured_heights: the ones above) */
    const float *base_ang_vel, *dof_pos, *default_dof_pos, *dof_vel, *actions, *forces;
    int64_t ld_forces;
    const float *height_noise_offset, *u_obs, *noise_scale_vec, *u_heights;
    float *obs_buf, *privileged_obs_buf, *heights;
} DtcEnvStep;
int dtc_env_post_physics(const DtcEnvStep* st, const DtcGridCfg* grid, const DtcObsCfg* obs, int N, void* stream);

/* ---- rollout-side store (row f2) ---------------------------------------------------------
 * RolloutStorage.add_transitions, rsl_rl/rsl_rl/storage/rollout_storage.py:99-116: the 13 `copy_` of one env step as
 * ONE launch (each item: N rows of width_bytes, source row stride src_stride_bytes -- 0 for a broadcast row such as
 * action_sigma -- into a dense destination), plus the time-out bootstrap of PPO.process_env_step, ppo.py:162-163:
 * rewards_dst[n] = rewards[n] + gamma * values[n] * time_outs[n] (time_outs NULL = plain copy; rewards_dst NULL = skip).
 * `items` is a HOST array of at most 16 descriptors holding DEVICE pointers. */
typedef struct DtcRowCopy {
    const void* src;
    void* dst;
    int64_t src_stride_bytes;
    int32_t width_bytes;
} DtcRowCopy;
int dtc_store_transition(const DtcRowCopy* items, int n_items, const float* rewards, const float* values,
                         const uint8_t* time_outs_or_null, float gamma, float* rewards_dst, int N, void* stream);

/* HistoryWrapper.step / get_observations, rsl_rl/rsl_rl/env/wrappers/history_wrapper.py:23,37:
 * out = cat(obs_history[:, num_obs:], obs) ([N, history_len*num_obs] <= 1024 floats per row); `out` may alias
 * `obs_history`.  The reference builds a NEW tensor every step and callers keep the old one until the transition is
 * stored, so the host wrapper ping-pongs two buffers.  Rows with reset[n] != 0 are cleared first (reset_idx, :42). */
int dtc_history_roll(const float* obs_history, const float* obs, float* out, const uint8_t* reset_or_null, int N,
                     int history_len, int num_obs, void* stream);

/* ---- RolloutStorage.compute_returns: rsl_rl/rsl_rl/storage/rollout_storage.py:138-152 --- */
/* GAE scan; writes returns and the UN-normalised advantages (returns - values) and
 * stats[0] = sum(advantages) (double).  stats is double[4] on the device. */
int dtc_gae(const float* rewards, const float* values, const uint8_t* dones, const float* last_values,
            float gamma, float lam, float* returns, float* advantages, double* stats, int T, int N,
            void* stream);
/* stats[1] = sum((adv - stats[0]/count)^2).  `count` is the GLOBAL sample count (data-parallel
 * callers all-reduce stats[0] first and pass world_size*T*N). */
int dtc_adv_sqdev(const float* advantages, double* stats, int64_t n_local, double count, void* stream);
/* adv = (adv - mean) / (std_unbiased + 1e-8) with mean = stats[0]/count, std from stats[1]. */
int dtc_adv_normalize(float* advantages, const double* stats, int64_t n_local, double count, void* stream);

/* ---- mini-batch gather: rollout_storage.py:195-209 (`tensor[batch_idx]`) ------------------ */
int dtc_gather_rows(const void* src, const int64_t* idx, void* dst, int64_t rows, int64_t row_bytes,
                    void* stream);
/* inverse map, dst[idx[r]] = src[r] (fp32 rows): the backward of `unpad_trajectories`
 * (rsl_rl/rsl_rl/utils/utils.py:67-70) -- gradients of the valid steps go back into the padded layout. */
int dtc_scatter_rows(const float* src, const int64_t* idx, float* dst, int64_t rows, int64_t row_floats,
                     void* stream);

/* ---- dense layers (nn.Linear + ReLU/ELU and their autograd; actor_critic_decoder.py:98-188,
 *      323-349).  A "segmented matrix" is the virtual concatenation torch.cat([...], dim=1) of
 *      up to 4 column blocks, each optionally row-gathered by the mini-batch index
 *      (ppo.py:201, actor_critic_decoder.py:431,550; rollout_storage.py:195-209) -- the cat and
 *      the gather are never materialised. ------------------------------------------------- */
typedef struct DtcSeg {
    float* ptr;          /* base of the source matrix (NULL = segment absent for outputs)       */
    int64_t ld;          /* leading dimension of the source matrix                              */
    int32_t col0;        /* first column inside the source matrix                               */
    int32_t width;       /* number of columns of this block                                     */
    int32_t gather;      /* 1: row m of the virtual matrix is row idx[m] of the source          */
    int32_t accumulate;  /* outputs only: 1 = add into the destination instead of overwriting   */
    int64_t rows;        /* inputs: number of rows of the source matrix (rows*ld floats behind ptr are readable: the
                          * loaders read 16 bytes at a time and bound their buffer descriptor with it -- reads past
                          * the last row return 0); outputs: unused                                              */
    uint32_t* amax;      /* two-term fp16 path (dtc_*_h2) only, device pointer.  Inputs: slot that holds the bit pattern of the
                          * largest |x| of the source tensor (or of any superset of the block: an upper bound costs precision only
                          * when it is more than ~2^10 too large), written by dtc_amax or by the kernel that produced the tensor.
                          * Outputs: slot that receives max(slot, largest |value| this call writes to the block), or NULL    */
} DtcSeg;

typedef struct DtcSegMat {
    int32_t nseg;        /* 1..4                                                                */
    int32_t cols;        /* sum of widths                                                       */
    const int64_t* idx;  /* mini-batch row indices for gathered segments (may be NULL)          */
    DtcSeg seg[4];
} DtcSegMat;

/* dst[rows, X->cols] (row stride ld_dst) = the segments of X side by side, row-gathered where a segment asks for it: packs
 * the narrow leading blocks of a layer input into one dense operand (see csrc/gae.hip: dtc_pack_cols). */
int dtc_pack_cols(const DtcSegMat* X, float* dst, int64_t ld_dst, int64_t rows, uint32_t* dst_amax /* amax slot of dst (two-term fp16 GEMM
                  path, see DtcSeg.amax) or NULL */, void* stream);

/* Y[M,N] = act(X[M,K] W[N,K]^T + b).  X is segmented (host struct). */
int dtc_linear_fwd(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy,
                   int M, int N, int K, int act, void* stream);
/* Same layer with the ReLU activation, additionally writing the SIGN RECORD of its output: one bit per element
 * (y > 0), dtc_relu_mask_elems(M, N) 16-bit words -- word [(row / 32) * 2 + half][col] holds the 16 rows
 * 4 * half + (r & 3) + 8 * (r >> 2), r = bit index, of its 32-row block (the accumulator layout of the kernels).
 * dtc_linear_dgrad_mask reads it instead of the saved activation: 1/32 of the bytes, the same bits as
 * `y > 0 ? g : 0` (torch's ReLU backward, rsl_rl/modules/actor_critic_decoder.py:98-131 encoder / decoder stacks under
 * ppo.py:252, 333).  Requires M % 128 == 0 and N a multiple of the launch's tile width (N % 128 == 0, or N == 64). */
int64_t dtc_relu_mask_elems(int M, int N);
int dtc_linear_fwd_mask(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, uint16_t* relu_mask,
                        int M, int N, int K, void* stream);
/* dX[M,K] = (dZ[M,N] W[N,K]) * (relu_mask bit), single-segment destination; M % 128 == 0, K as N above. */
int dtc_linear_dgrad_mask(const float* dZ, int64_t lddz, const float* W, const DtcSegMat* dX, const uint16_t* relu_mask,
                          int M, int N, int K, void* stream);
/* dtc_linear_fwd / dtc_linear_fwd_mask (relu_mask may be NULL) that also adds the largest |Y| it writes to the amax record y_amax of
 * the two-term fp16 GEMM path (DtcSeg.amax below) -- for the narrow layers whose result a split-path kernel consumes.  dtc_linear_dgrad /
 * dtc_linear_dgrad_mask do the same for a SINGLE destination block that brings a record in dX->seg[0].amax. */
int dtc_linear_fwd_amax(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, uint16_t* relu_mask,
                        uint32_t* y_amax, int M, int N, int K, int act, void* stream);
/* ---- split-precision path (csrc/gemm_s3.hip): the same products on the bf16 matrix pipe, every fp32 operand split into
 * three bf16 terms (a = a1 + a2 + a3 exactly to 2^-24 |a|) and the six leading cross products accumulated in fp32 -- fp32-level
 * accuracy (tests/test_hip_split.py measures it against fp64 next to the single-pass kernels) at up to 2.67x the fp32 MFMA
 * rate; results are NOT bit-identical to the fmaf chain of dtc_linear_fwd.  Same argument meaning as dtc_linear_fwd /
 * dtc_linear_fwd_mask (relu_mask may be NULL) and dtc_linear_dgrad / dtc_linear_dgrad_mask. */
/* Library-side switch for the entry points that have no `_s3` twin of their own: with it on, dtc_linear_wgrad (and through it
 * the W_hh / W_ih weight gradients inside dtc_gru_bwd / dtc_lstm_bwd) runs as a one-layer dtc_wgrad_group_s3, and
 * dtc_linear_wgrad_workspace / dtc_gru_workspace / dtc_lstm_workspace return sizes that fit either path.  Off by default;
 * dtc_amd/ops.py sets it from DTC_GEMM_SPLIT. */
void dtc_set_gemm_split(int on);
int dtc_get_gemm_split(void);
/* Weight image: the W operand of a split-path call as the kernel's LDS planes -- for every 128-column tile and 16-k stage one
 * 12 KiB chunk [plane 3][row 128][16 k bf16], copied into LDS by LDS-DMA (no conversion work for W inside the K loop).  Built by
 * the call itself into `wplanes` (wimage_ready = 0; scratch of dtc_s3_planes_bytes(N, K) bytes -- for the data gradient
 * (K, N) --, 16-byte aligned, private to the call until it has completed on its stream), or beforehand for many calls at once
 * by dtc_s3_wimage_group (wimage_ready = 1: `wplanes` is the image of exactly this call: same W, shape and operand segments). */
int64_t dtc_s3_planes_bytes(int N, int K);
typedef struct DtcWimgJob {
    const float* W;              /* [N, K] as stored                                                                        */
    void* img;                   /* dtc_s3_planes_bytes(N, K) (trans: (K, N)) bytes                                         */
    const DtcSegMat* seg;        /* trans == 0: the X operand of the forward call; trans == 1: the dX destination          */
    int32_t N, K, trans;
} DtcWimgJob;
int dtc_s3_wimage_group(const DtcWimgJob* jobs, int count, void* stream);
int dtc_linear_fwd_s3(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, uint16_t* relu_mask,
                      void* wplanes, int wimage_ready, int M, int N, int K, int act, void* stream);
int dtc_linear_dgrad_s3(const float* dZ, int64_t lddz, const float* W, const DtcSegMat* dX, const float* Xsaved,
                        int64_t ldxs, const uint16_t* relu_mask, void* wplanes, int wimage_ready, int M, int N, int K, int act,
                        void* stream);
/* dtc_linear_fwd_mse / dtc_linear_fwd_mse_parts on the split-precision path */
/* ---- block-scaled two-term fp16 operand images (round 5; csrc/h2i_core.hpp, gemm_h2i.hip, wgrad_h2i.hip): THE representation of the
 * wide layers' operands -- the fp32 products of actor_critic_decoder.py:98-188, 323-349 (under ppo.py:197-218, 252, 265, 289, 333) with
 * every operand held in HBM as the (hi, lo) fp16 planes the K loops read by LDS-DMA, 4 bytes per element, scaled by a power of two chosen
 * PER ROW and block of 128 columns (weights: per 128 x 128 block) from the block's own largest finite element:
 *   image(M, K) = ceil(M / 128) x ceil(K / 16) chunks of 8 KiB [plane 2][slot 256][16 bytes] + int32 exps[row tile][k block][128];
 *   dtc_h2i_bytes(M, K) bytes, 16-byte aligned, < 2 GiB.  Rows >= M and columns >= K are zero.
 * Every element is exact to 2^-22 of its row block's largest element whatever the rest of the tensor holds (no tensor-wide amax, no
 * records, nothing to keep in step on the host), and a non-finite element poisons only the outputs that depend on it -- its row in the
 * forward / data-gradient products, its column of dW in the weight gradient -- as in the fp32 reference.
 * Producers: dtc_h2i_pack (fp32 segmented / row-gathered operand -> image: the rollout storage, narrow hand-over tensors), the epilogues
 * of dtc_linear_fwd_h2i / _mse_h2i / dtc_linear_dgrad_h2i (results as fp32, as an image, or both), dtc_h2i_wimage_group (weights).
 * dtc_h2i_unpack decodes an image (tests). */
int64_t dtc_h2i_bytes(int M, int K);
/* tuning: forward / data-gradient launches with at most max_tiles 128 x 128 result tiles run on 64-row tiles (twice the workgroups; default
 * 256 = one tile per CU; 0: never; -1: back to DTC_H2I_ROWS64_MAX / the default).  Results are bit-identical either way. */
void dtc_h2i_rows64_max(int max_tiles);
void dtc_h2i_trace(void* buf);   /* debug: per-workgroup {start, K loop done, end (100 MHz ticks), HW_ID} records of the launches that follow (NULL: off) */
int dtc_h2i_pack(const DtcSegMat* X, int M, void* img, void* stream);
int dtc_h2i_unpack(const void* img, int M, int K, float* out, int64_t ld, void* stream);
/* A weight image.  Rows: up to 2 ranges (r0, nr) of the operand's rows, one after the other, each padded to whole 128-row tiles (all but the
 * last must be multiples of 128 long).  Reduction: up to 4 column ranges (c0, cw) side by side, each padded to whole 16-column stages --
 * the walk of the row operand's images.  trans = 0: element (row, c) = W[row * ld + c] (forward: rows (0, N), ranges = the column blocks of
 * W that meet the row operand's images, in that order); trans = 1: W[c * ld + row] (data gradient: rows = windows of W's columns, one
 * reduction range (0, N)).  img: dtc_h2i_wimage_bytes(job) bytes. */
typedef struct DtcH2iWJob {
    const float* W;
    int64_t ld;
    void* img;
    int32_t trans, nrows, nseg;
    int32_t r0[2], nr[2];
    int32_t c0[4], cw[4];
} DtcH2iWJob;
int64_t dtc_h2i_wimage_bytes(const DtcH2iWJob* job);
int dtc_h2i_wimage_group(const DtcH2iWJob* jobs, int count, void* stream);
/* the row operand of a product: up to 4 images over the same M rows, side by side along the reduction (at most 2048 columns in all) */
typedef struct DtcH2iOperand {
    int32_t nseg;
    int32_t width[4];
    const void* img[4];
} DtcH2iOperand;
/* Y = act(X W^T + b) (actor_critic_decoder.py:98-131, 323-349); results: fp32 Y [M, >= N] (may be NULL) and / or Yimg = image(M, N) */
int dtc_linear_fwd_h2i(const DtcH2iOperand* X, const void* wimg, const float* b, float* Y, int64_t ldy, void* Yimg, uint16_t* relu_mask,
                       int M, int N, int act, void* stream);
/* Chains of narrow layers in ONE launch (csrc/gemm_h2i.hip: chain_h2i_kernel): up to three layers of at most 512 result columns each (round 6; a layer wider than
 * 128 columns = several column tiles, run one after the other by the row tile's workgroup: the actor's / critic's tails 512 -> 256 -> 128,
 * actor_critic_decoder.py:323-349, forward and backward),
 * layer i + 1 reading the image layer i wrote (the workgroup that owns a row tile runs the layers one after the other on it).  Same
 * arguments, same results bit for bit as the per-layer calls: the CE-net encoder 265 -> 128 -> 64 -> 35 / decoder 531 -> 64 -> 128 -> 53
 * (actor_critic_decoder.py:98-142) forward, and the data-gradient chains back through them. */
typedef struct DtcH2iFwdLayer {
    DtcH2iOperand X;             /* layer 0: up to 4 images; layers > 0: exactly the previous layer's Yimg                    */
    const void* wimg;
    const float* b;
    float* Y;                    /* fp32 result (may be NULL)                                                                   */
    int64_t ldy;
    void* Yimg;                  /* image result (NULL only for the last layer)                                                 */
    uint16_t* relu_mask;
    int32_t N, act;
} DtcH2iFwdLayer;
int dtc_linear_fwd_chain_h2i(const DtcH2iFwdLayer* layers, int count, int M, void* stream);
typedef struct DtcH2iDgradLayer {
    const void* dZimg;           /* layer 0: image(M, N); layers > 0: the previous layer's dXimg                                */
    const void* wimgT;
    const DtcSegMat* dX;         /* fp32 destination blocks (may be NULL)                                                       */
    void* dXimg;                 /* image result (NULL only for the last layer)                                                 */
    const float* add;
    int64_t ld_add;
    const float* Xsaved;
    int64_t ldxs;
    const uint16_t* relu_mask;
    int32_t N, Kwin, img_cols, act;
} DtcH2iDgradLayer;
int dtc_linear_dgrad_chain_h2i(const DtcH2iDgradLayer* layers, int count, int M, void* stream);
/* the terrain decoder's output layer fused with its MSE (ppo.py:223): dL/dY as fp32 (may be NULL) and / or image(M, N) */
int64_t dtc_linear_fwd_mse_h2i_parts(int M, int N);
int dtc_linear_fwd_mse_h2i(const DtcH2iOperand* X, const void* wimg, const float* b, const float* target, int64_t ldt, int64_t target_rows,
                           int tcol0, const int64_t* tidx, float scale, float* dY, int64_t lddy, void* dYimg, double* sq_part, int M, int N,
                           void* stream);
/* dX[:, window] = ((dZ W[:, window]) + add) * act'(.): dZimg = image(M, N), wimgT = image of W^T for the window (Kwin columns: the job's
 * row ranges one after the other); results over the window: fp32 destination blocks dX (may be NULL) and / or dXimg = image(M, img_cols)
 * of the window's first img_cols columns (0: all Kwin); add: fp32 [M, ld_add] or NULL */
int dtc_linear_dgrad_h2i(const void* dZimg, int N, const void* wimgT, int Kwin, const DtcSegMat* dX, void* dXimg, int img_cols,
                         const float* add, int64_t ld_add, const float* Xsaved, int64_t ldxs, const uint16_t* relu_mask, int M, int act,
                         void* stream);
/* dW_j = dZ_j^T X_j, db_j = colsum(dZ_j) for `count` <= 12 layers in ONE launch, both operands images over the same M batch rows
 * (M a multiple of 128 per batch slice: the per-row exponents of the two operands are folded into one operand's fragments per batch
 * row, csrc/wgrad_h2i.hip).  workspace >= dtc_wgrad_group_h2i_workspace() bytes, 16-byte aligned. */
typedef struct DtcWgradH2iJob {
    const void* dZimg;   /* image(M, N) */
    const void* Ximg;    /* image(M, K) */
    float* dW;           /* [N, ldw >= K]: columns [wcol0, wcol0 + K) of the layer's weight gradient */
    float* db;           /* [N] or NULL   */
    int64_t ldw;
    int32_t N, K, wcol0;
} DtcWgradH2iJob;
int64_t dtc_wgrad_group_h2i_workspace(const DtcWgradH2iJob* jobs, int count, int M);
int dtc_wgrad_group_h2i(const DtcWgradH2iJob* jobs, int count, int M, void* workspace, void* stream);
/* ---- two-term fp16 path (round 4, csrc/s3_core.hpp): the same products with every fp32 operand x scaled by a power of two chosen from
 * its tensor's amax (largest |x|: scaled into [2^14, 2^15)) and written as hi + lo, hi = fp16(x 2^e), lo = fp16(x 2^e - hi): 22 significant
 * bits (elements more than 2^18 below amax: absolute error 2^-40 amax), THREE fp16 MFMA passes per product (lo hi', hi lo', hi hi'; exact
 * in the fp32 accumulator) instead of six bf16 ones, the sums scaled back exactly.  Error level of the fp32 MFMA chain
 * (tests/test_hip_split.py); half the matrix-pipe work and energy of the bf16 x 3 path.  An operand brings its amax in a device
 * SLOT -- a record of dtc_amax_record_bytes() bytes (16 words on 16 cache lines: the waves of a producer spread their atomic maxima over
 * them, readers take the largest; each word the BIT PATTERN of a float; unsigned order = float order, NaN above everything) --:
 * DtcSeg.amax for segmented operands, dz_amax for dZ; NULL = the operand has none and the call computes it (one memset + one launch in
 * front of the GEMM for all such operands of the call).  A producer adds its result to a slot: ZERO it before the first kernel that writes the tensor (a stale
 * larger value is harmless up to ~2^10; a smaller one overflows fp16 and the result turns NaN -- never silently wrong).  Weight
 * images: same jobs and buffers as the bf16 ones, built by dtc_h2_wimage_group or by the call (wimage_ready = 0); the weight's own
 * amax lives inside the image buffer. */
int64_t dtc_amax_record_bytes(void);                                        /* size of one slot (see below)                   */
int dtc_amax(const DtcSegMat* X, int M, uint32_t* slot, void* stream);      /* slot = amax over rows < M of all segments of X */
int dtc_h2_wimage_group(const DtcWimgJob* jobs, int count, void* stream);
int dtc_linear_fwd_h2(const DtcSegMat* X, const float* W, const float* b, float* Y, int64_t ldy, uint16_t* relu_mask,
                      void* wplanes, int wimage_ready, uint32_t* y_amax, int M, int N, int K, int act, void* stream);
int dtc_linear_dgrad_h2(const float* dZ, int64_t lddz, const uint32_t* dz_amax, const float* W, const DtcSegMat* dX, const float* Xsaved,
                        int64_t ldxs, const uint16_t* relu_mask, void* wplanes, int wimage_ready, int M, int N, int K, int act,
                        void* stream);
int dtc_linear_fwd_mse_h2(const DtcSegMat* X, const float* W, const float* b, const float* target, int64_t ldt,
                          int64_t target_rows, int tcol0, const int64_t* tidx, float scale, float* dY, int64_t lddy,
                          double* sq_part, void* wplanes, int wimage_ready, uint32_t* dy_amax, int M, int N, int K, void* stream);
/* dtc_wgrad_group_s3 on the fp16 path: every job brings dz_amax and X.seg[i].amax (same workspace size) */
int dtc_wgrad_group_h2(const struct DtcWgradJob* jobs, int count, int M, void* workspace, void* stream);
int64_t dtc_linear_fwd_mse_s3_parts(int M, int N);
int dtc_linear_fwd_mse_s3(const DtcSegMat* X, const float* W, const float* b, const float* target, int64_t ldt,
                          int64_t target_rows, int tcol0, const int64_t* tidx, float scale, float* dY, int64_t lddy,
                          double* sq_part, void* wplanes, int wimage_ready, int M, int N, int K, void* stream);
/* GRU recurrence on the split-precision path (csrc/gru_s3.hip; H a multiple of 128): the per-time-step products with W_hh as an
 * LDS image built once per pass (dtc_gru_s3_image: backward = 0 for the forward steps, 1 for the data-gradient chunks), row
 * operands loaded two stages ahead -- kernels for the ~1500-row, one-workgroup-per-CU launches of one time step.  dtc_gru_fwd /
 * dtc_gru_bwd use them for T >= 4 when dtc_get_gemm_split() is on (DTC_GRU_S3=0: single-pass kernels). */
int64_t dtc_gru_s3_image_bytes(int H);
int dtc_gru_s3_image(const float* W_hh, void* img, int H, int backward, void* stream);
int dtc_gru_step_fwd_s3(const float* hprev, const void* img, const float* b_hh, const float* gi_t, float* hout, float* gates_t,
                        float* hn_t, int R, int H, void* stream);
/* chunk c of dgh_t [R, 3H] W_hh [3H, H] -> part + c * part_stride ([R, H]); (3H / nparts) a multiple of 32 */
int dtc_gru_dgrad_parts_s3(const float* dgh_t, const void* img, float* part, int64_t part_stride, int R, int H, int nparts,
                           void* stream);
/* dtc_wgrad_group on the split-precision path (same jobs, same outputs; its own workspace size). */
int64_t dtc_wgrad_group_s3_workspace(const struct DtcWgradJob* jobs, int count, int M);
int dtc_wgrad_group_s3(const struct DtcWgradJob* jobs, int count, int M, void* workspace, void* stream);
/* A chain of forward layers in ONE call (same kernels, same results as `count` dtc_linear_fwd calls issued in order on
 * `stream`): the rollout side of PPO.act / evaluate (ppo.py:137-155) is launch-bound -- ~16 small layers per env step at
 * M = num_envs rows -- and the host cost of marshalling each layer through the FFI is paid once per chain instead of
 * once per layer.  The caller may keep the array and patch input pointers between steps. */
typedef struct DtcFwdLayer {
    DtcSegMat X;         /* [M, K] segmented input                                              */
    const float* W;      /* [N, K]                                                               */
    const float* b;      /* [N] or NULL                                                          */
    float* Y;            /* [M, >= N]                                                            */
    int64_t ldy;
    int32_t N, K, act;
} DtcFwdLayer;
int dtc_linear_fwd_list(const DtcFwdLayer* layers, int count, int M, void* stream);

/* dX[M,K] = (dZ[M,N] W[N,K]) * act'(Xsaved), written through the segmented destination dX
 * (segments with ptr == NULL are skipped).  Xsaved (post-activation output of the previous
 * layer, leading dimension ldxs) may be NULL when act == DTC_ACT_NONE. */
int dtc_linear_dgrad(const float* dZ, int64_t lddz, const float* W, const DtcSegMat* dX,
                     const float* Xsaved, int64_t ldxs, int M, int N, int K, int act, void* stream);
/* Data gradient with the reduction index split into `nsplit` equal chunks that run side by side in ONE launch
 * (for GEMMs with few rows, e.g. one GRU time step: M ~ 1500, N = 3H): chunk g computes
 * dX + g*split_stride [M,K] (row stride lddx) = dZ[:, g*N/nsplit : (g+1)*N/nsplit] W[g*N/nsplit : (g+1)*N/nsplit, :];
 * no activation, no accumulation -- the caller adds the chunks in a fixed order (deterministic). */
int dtc_linear_dgrad_split(const float* dZ, int64_t lddz, const float* W, float* dX, int64_t lddx, int64_t split_stride,
                           int M, int N, int K, int nsplit, void* stream);

/* Host hint: non-zero while the caller launches weight gradients on a second stream next to the stream of the data
 * gradients (the overlapped schedule of PPO.update).  The data-gradient kernel then leaves one of its six workgroup slots
 * per CU to them.  Process-wide, not thread-safe against concurrent launches; results never depend on it. */
void dtc_set_concurrency_hint(int side_stream_active);

/* dW[N,K] = dZ^T X, db[N] = column sums of dZ.  workspace: >= dtc_linear_wgrad_workspace() bytes. */
int64_t dtc_linear_wgrad_workspace(int M, int N, int K);
int dtc_linear_wgrad(const float* dZ, int64_t lddz, const DtcSegMat* X, float* dW, float* db,
                     void* workspace, int M, int N, int K, void* stream);

/* The weight gradients of ALL layers of one gradient bucket (ppo.py:252 / 333: the parameters one `loss.backward()`
 * reaches; the decoders or the encoders of the VAE step, the actor + critic bodies or the encoders of the PPO step)
 * in ONE partial-product launch + ONE reduce launch.  Every job shares the row count M (the mini-batch) and runs
 * exactly as dtc_linear_wgrad would (same segmented / gathered X, same outputs); count <= 12.
 * workspace >= dtc_wgrad_group_workspace() bytes, 16-byte aligned, private to the call until it has completed. */
typedef struct DtcWgradJob {
    const float* dZ;     /* [M, N], row stride lddz                                              */
    int64_t lddz;
    DtcSegMat X;         /* [M, K] segmented input of the layer                                  */
    float* dW;           /* [N, K]                                                               */
    float* db;           /* [N] or NULL                                                          */
    int32_t N, K;
    int64_t dz_rows;     /* 0: row m of the product is dZ[m].  > 0 (split path only): dZ has dz_rows rows and row m of the
                          * product is dZ[X.idx[m]] -- the SAME row map as X's gathered segments (the recurrent trainers' valid
                          * rows of the padded trajectory layout: the padding rows carry zero gradient and are skipped)      */
    const uint32_t* dz_amax;   /* dtc_wgrad_group_h2 only: amax slot of dZ (see DtcSeg.amax)                                   */
} DtcWgradJob;
int64_t dtc_wgrad_group_workspace(const DtcWgradJob* jobs, int count, int M);
/* dtc_linear_wgrad over the rows idx[0 .. M) of BOTH operands (dZ [dz_rows, N] and the single-segment X [x_rows, K] share the row
 * map; split-precision path only: returns DTC_ERR_ARG when it is switched off -- the caller then runs the padded product). */
int dtc_linear_wgrad_rows(const float* dZ, int64_t lddz, int64_t dz_rows, const float* X, int64_t ldx, int64_t x_rows,
                          const int64_t* idx, float* dW, float* db, void* workspace, int M, int N, int K, void* stream);
int dtc_wgrad_group(const DtcWgradJob* jobs, int count, int M, void* workspace, void* stream);

/* ---- CE-net latent: actor_critic_decoder.py:274-302 ---------------------------------------- */
/* mulv [B,35] = [latent_mu(19) | latent_var(16)].  In place: outlier -> lower median of the
 * non-outliers (2-sigma rule on the batch statistics), then z = eps*exp(0.5*lv) + mu[:,3:].
 * mask [B,16] uint8 receives the outlier mask, info (int32[4]) {n_outliers, median_flat_index,
 * median bits, 0}.  workspace >= dtc_cenet_workspace() bytes. */
int64_t dtc_cenet_workspace(int B);
int dtc_cenet_latent_fwd(float* mulv, const float* eps, float* z, uint8_t* mask, int32_t* info,
                         void* workspace, int B, uint32_t* z_amax /* amax record of z (two-term fp16 GEMM path, DtcSeg.amax) or NULL */,
                         void* stream);
/* ... and zmu_img (may be NULL): the operand image (dtc_h2i_bytes(B, 19)) of [z | mu[:, :3]] -- the narrow input block of the CE-net
 * decoder's and the actor's first layer (actor_critic_decoder.py:431, 300) -- written by the same launch instead of a pack launch. */
int dtc_cenet_latent_fwd_img(float* mulv, const float* eps, float* z, uint8_t* mask, int32_t* info, void* workspace, int B,
                             uint32_t* z_amax, void* zmu_img, void* stream);
/* backward of the above: dmulv [B,35] holds the direct gradients w.r.t. (mu, lv_fixed) on entry
 * and the gradients w.r.t. the raw head outputs on exit. */
int dtc_cenet_latent_bwd(float* dmulv, const float* dz, const float* eps, const float* mulv,
                         const uint8_t* mask, const int32_t* info, void* workspace, int B,
                         uint32_t* dmulv_amax /* amax record of the whole [B,35] gradient on exit, or NULL */, void* stream);

/* ---- losses (fused forward + gradient) ------------------------------------------------------ */
/* VAE losses of ppo.py:205-247.  Gathered operands are read as src[idx[b]].  Outputs:
 * d_recons [B,53], d_hrecon [B,693], dmulv [B,35] (vel + 4*kld parts) and
 * losses[0..3] = {recons, vel, kld, height} (float, device). */
int dtc_vae_loss(const float* recons, const float* hrecon, const float* mulv, const float* next_obs,
                 const float* priv, const float* base_vel, const int64_t* idx, float* d_recons,
                 float* d_hrecon, float* dmulv, float* losses, void* workspace, int B,
                 uint32_t* drec_amax /* amax record of d_recons or NULL */, void* stream);

/* The terrain-decoder output layer fused with its loss (ppo.py:216-218: height_recon = terrain_decoder(l_t),
 * height_loss = mse(height_recon, priv[..., 696:])): e = (X W^T + b) - target[tidx[m], tcol0 + n]; writes
 * dY[m,n] = e * scale (scale = 2/(M*N): dL/d height_recon) and one double per workgroup, sum(e^2), into sq_part
 * (dtc_linear_fwd_mse_parts(M, N) slots).  height_recon itself never reaches HBM.  dtc_vae_loss_fused is dtc_vae_loss
 * without the height term; it adds the partials (in index order) into losses[3]. */
int64_t dtc_linear_fwd_mse_parts(int M, int N);
int dtc_linear_fwd_mse(const DtcSegMat* X, const float* W, const float* b, const float* target /*[target_rows, ldt]*/,
                       int64_t ldt, int64_t target_rows, int tcol0, const int64_t* tidx /*[M]*/, float scale, float* dY,
                       int64_t lddy, double* sq_part, int M, int N, int K, void* stream);
int dtc_vae_loss_fused(const float* recons, const float* mulv, const float* next_obs, const float* base_vel,
                       const int64_t* idx, float* d_recons, float* dmulv, const double* height_sq_part, int n_height_part,
                       float* losses, void* workspace, int B, uint32_t* drec_amax /* as dtc_vae_loss */, void* stream);
/* ... and drec_img (may be NULL): the operand image (dtc_h2i_bytes(B, 53)) of d_recons, written by the same launch. */
int dtc_vae_loss_fused_img(const float* recons, const float* mulv, const float* next_obs, const float* base_vel,
                           const int64_t* idx, float* d_recons, float* dmulv, const double* height_sq_part, int n_height_part,
                           float* losses, void* workspace, int B, uint32_t* drec_amax, void* drec_img, void* stream);
int64_t dtc_loss_workspace(int B);

typedef struct DtcPpoCfg {
    float clip_param, value_loss_coef, entropy_coef, desired_kl;
    int32_t use_clipped_value_loss, adaptive_schedule;
    float* kl_mirror;      /* NULL, or a device address that ALSO receives kl_mean from the finalize launch (data parallel: slot 0 of the
                              gradient header the first bucket's all-reduce carries -- no copy of its own between the loss and the backward pass) */
} DtcPpoCfg;
/* PPO losses + KL-adaptive learning rate of ppo.py:288-327.  mean [B,12], std [12], value [B].
 * Outputs: dmean [B,12], dvalue [B], dstd [12], losses[0..3] = {surrogate, value, entropy, kl_mean},
 * lr (double, device; read-modify-write exactly as ppo.py:301-307). */
int dtc_ppo_loss(const float* mean, const float* std, const float* value, const float* actions,
                 const float* old_logp, const float* old_mu, const float* old_sigma, const float* advantages,
                 const float* returns, const float* old_values, const int64_t* idx, const DtcPpoCfg* cfg,
                 float* dmean, float* dvalue, float* dstd, float* losses, double* lr, void* workspace,
                 int B, int num_actions, void* stream);
/* The actor's and the critic's output layers (mean = Ha Wa^T + ba, value = Hc Wc^T + bc over the last hidden activations
 * Ha, Hc [B,H], H in {64, 128, 256}), dtc_ppo_loss on them, and the data gradients of the two layers
 * (dHa = (dmean Wa) * act'(Ha), dHc = (dvalue Wc) * act'(Hc), act' through the saved post-activation values as in
 * dtc_linear_dgrad) in ONE launch + the finalize launch: replaces two dtc_linear_fwd, dtc_ppo_loss and two
 * dtc_linear_dgrad calls on the critical path of every policy step (ppo.py:288-327 with actor_critic_decoder.py:
 * 330-345).  Also writes mean [B,A], value [B], dmean, dvalue (the weight gradients of the two layers need them). */
int dtc_ppo_heads_loss(const float* Ha, int64_t ldha, const float* Hc, int64_t ldhc, int H, const float* Wa /*[A,H]*/,
                       const float* ba, const float* Wc /*[1,H]*/, const float* bc, int act_prev, const float* std,
                       const float* actions, const float* old_logp, const float* old_mu, const float* old_sigma,
                       const float* advantages, const float* returns, const float* old_values, const int64_t* idx,
                       const DtcPpoCfg* cfg, float* mean, float* value, float* dmean, float* dvalue, float* dHa,
                       int64_t lddha, float* dHc, int64_t lddhc, float* dstd, float* losses, double* lr, void* workspace,
                       int B, int num_actions, uint32_t* dha_amax, uint32_t* dhc_amax, uint32_t* dmean_amax,
                       uint32_t* dval_amax /* amax records of dHa / dHc / dmean / dvalue (two-term fp16 GEMM path, see DtcSeg.amax), each
                       may be NULL */, void* stream);
/* ... and the operand images (dtc_h2i_bytes(B, H), (B, H), (B, A), (B, 1); each may be NULL; the first two need H <= 128) of dHa, dHc, dmean,
 * dvalue, written by the same launch: the trainer's image-operand data / weight gradients read them without a pack launch.  Where an
 * image of dHa / dHc is given, its fp32 destination may be NULL (the image is then the only copy: 25 MB less to write per call). */
int dtc_ppo_heads_loss_img(const float* Ha, int64_t ldha, const float* Hc, int64_t ldhc, int H, const float* Wa, const float* ba,
                           const float* Wc, const float* bc, int act_prev, const float* std, const float* actions,
                           const float* old_logp, const float* old_mu, const float* old_sigma, const float* advantages,
                           const float* returns, const float* old_values, const int64_t* idx, const DtcPpoCfg* cfg, float* mean,
                           float* value, float* dmean, float* dvalue, float* dHa, int64_t lddha, float* dHc, int64_t lddhc,
                           float* dstd, float* losses, double* lr, void* workspace, int B, int num_actions, uint32_t* dha_amax,
                           uint32_t* dhc_amax, uint32_t* dmean_amax, uint32_t* dval_amax, void* dHa_img, void* dHc_img,
                           void* dmean_img, void* dval_img, void* stream);
/* The learning-rate rule of ppo.py:301-307 alone (data-parallel callers run dtc_ppo_loss with
 * adaptive_schedule = 0, all-reduce the KL mean, then call this so every rank takes the same branch).
 * The slot is CONSUMED: it is overwritten with a NaN of a payload of its own (0x7fc0dead), and finding exactly that pattern (a caller that
 * exchanged the gradient header without depositing this step's KL first) poisons *lr with NaN instead of silently re-using a stale value.
 * A KL that is NaN itself (any other payload) leaves *lr unchanged, as both comparisons of the reference do. */
int dtc_lr_adapt(float* kl_mean, double* lr, float desired_kl, float* kl_out /* NULL, or where the (averaged) KL is kept: the step's statistics row */, void* stream);
/* log-prob / sampling side of PPO.act (ppo.py:137-150): actions = mean + std*noise,
 * logp = sum_j log N(a; mean, std). */
int dtc_gaussian_act(const float* mean, const float* std, const float* noise, float* actions,
                     float* logp, float* mu_out, float* sigma_out, int B, int num_actions, void* stream);

/* ActorCriticDecoder.adapt_bootstrap_probability (actor_critic_decoder.py:404-407): out[0] = 1 - tanh(std(rewards) / mean(rewards)),
 * unbiased std as torch.std; rewards: n contiguous floats on the device, out: one device float. */
int dtc_bootstrap_probability(const float* rewards, int64_t n, float* out, void* stream);

/* ---- clip_grad_norm_ + Adam over one flat parameter range (ppo.py:253-254, 334-335) --------- */
/* gnorm_out (float, device) receives the pre-clip global L2 norm.  lr is a device double;
 * step is the 1-based Adam step count of this range. */
int dtc_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float max_grad_norm, const double* lr, double beta1, double beta2, double eps, int64_t step,
                  float* gnorm_out, void* workspace, void* stream);
int64_t dtc_adam_workspace(int64_t n);

/* ---- device random draws of PPO.update (counter-based: deterministic in (seed, offset / n), no host synchronisation) ----
 * dtc_randn: n standard-normal floats (Philox4x32-10 + Box-Muller; element 4g..4g+3 from counter g + offset) -- the
 * `torch.randn_like` of the CE-net reparameterisation (actor_critic_decoder.py:283).
 * dtc_randperm: a pseudo-random permutation of [0, n) as int64 (keyed Feistel bijection of [0, 2^k) with cycle walking; no
 * sort) -- the `torch.randperm` of the mini-batch generator (rollout_storage.py:165). */
int dtc_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
int dtc_randperm(int64_t* out, int64_t n, uint64_t seed, void* stream);

/* ---- GRU (torch.nn.GRU, 1 layer, gate order r,z,n; actor_critic_recurrent.py:92-116 `Memory`) ---- */
/* One fused GRU time step (forward), the building block of dtc_gru_fwd: the recurrent GEMM
 * gh = hprev W_hh^T + b_hh and the gate math of torch.nn.GRU in its epilogue (gh never reaches HBM):
 * hout [R,H], gates_t [R,3H] = (r | z | n), hn_t [R,H] = gh_n; gi_t [R,3H] = x_t W_ih^T + b_ih.  H % 32 == 0. */
int dtc_gru_step_fwd(const float* hprev /*[R,H]*/, const float* W_hh /*[3H,H]*/, const float* b_hh, const float* gi_t,
                     float* hout, float* gates_t, float* hn_t, int R, int H, void* stream);

/* gi [T,R,3H] = x W_ih^T + b_ih is computed by dtc_linear_fwd over all T*R rows; this runs the recurrence
 *     r = sig(gi_r + gh_r), z = sig(gi_z + gh_z), n = tanh(gi_n + r*gh_n), h_t = (1-z)*n + z*h_{t-1},
 *     gh = h_{t-1} W_hh^T + b_hh
 * for t < T over R sequences (R = padded trajectories of utils.py:33-64, or envs during the rollout).
 * hs_all is [T+1,R,H]: slot 0 receives h0, slot t+1 = h_t (so hs_all[1:] is nn.GRU's output and hs_all[:T]
 * the h_{t-1} operand of the backward pass).  gates [T,R,3H] receives (r, z, n), hn [T,R,H] = gh_n.
 * workspace >= dtc_gru_workspace() bytes. */
int64_t dtc_gru_workspace(int T, int R, int H);
int dtc_gru_fwd(const float* gi, const float* h0, const float* W_hh, const float* b_hh, float* hs_all,
                float* gates, float* hn, void* workspace, int T, int R, int H, void* stream);
/* BPTT.  dhs [T,R,H] = gradient w.r.t. the outputs h_1..h_T.  Produces dgi [T,R,3H] (gradient w.r.t. gi:
 * feed it to dtc_linear_wgrad with x for W_ih / b_ih), dW_hh [3H,H], db_hh [3H] and dh0 [R,H].
 * valid_rows (optional, n_valid entries of t * R + r): the (t, r) slots that belong to a trajectory -- the padding slots of the
 * padded layout carry zero gradient, and on the split-precision path the W_hh weight gradient then skips them.
 * dW_hh == db_hh == NULL: the W_hh weight gradient is left to the caller, who finds dgh_all [T,R,3H] (gradient w.r.t. the recurrent
 * pre-activations gh_t) at workspace + dtc_gru_dgh_offset(T, R, H) once the call has run -- dW_hh = dgh_all^T hs_all[:T], db_hh =
 * colsum(dgh_all); the operand-image trainers pack both into images and add the product to their grouped weight-gradient launch. */
int64_t dtc_gru_dgh_offset(int T, int R, int H);
int dtc_gru_bwd(const float* dhs, const float* hs_all, const float* gates, const float* hn, const float* W_hh,
                float* dgi, float* dW_hh, float* db_hh, float* dh0, void* workspace, const int64_t* valid_rows, int n_valid,
                int T, int R, int H, void* stream);

/* The recurrence as ONE persistent launch over all T time steps (csrc/gru_seq.hip; H = 512, R <= 2048, T >= 2): a workgroup owns a row
 * block x 16 hidden units, keeps its slice of W_hh in LDS as two-term fp16 for all steps and meets the other workgroups of its row block at
 * a counter barrier once per step.  OPT-IN (DTC_GRU_SEQ=1 or dtc_set_gru_seq(1); -1 = back to the environment's choice): dtc_gru_fwd / dtc_gru_fwd_multi then use it
 * on the split-precision path whenever dtc_gru_seq_supported(); results within fp32 rounding of the per-step kernels (another, equally
 * accurate, split of the operands).  Default: the per-step launches (measured no slower in the trainers).  A launch whose workgroups cannot meet (more than two such launches sharing the device) gives up after 2 s and raises the
 * flag dtc_gru_seq_status() returns (1; it synchronises with the device; reset != 0 clears it) -- its outputs are then incomplete.
 * seq_ws: dtc_gru_seq_workspace(R, H) bytes, 16-byte aligned (dtc_gru_workspace() includes them). */
void dtc_set_gru_seq(int on);
int dtc_get_gru_seq(void);
void dtc_gru_seq_trace(void* buf);   /* debugging: (workgroups x T x 4) 8-byte time stamps (100 MHz) per step of thread 0: met / K loop done / gates done / arrived */
int dtc_gru_seq_supported(int T, int R, int H, int exclusive /* 0: a launch that leaves half the CUs to a second one (dtc_gru_fwd); 1: dtc_gru_fwd_multi's */);
int dtc_gru_seq_status(int reset);
int64_t dtc_gru_seq_workspace(int R, int H);
int dtc_gru_seq_fwd(const float* gi, const float* h0, const float* W_hh, const float* b_hh, float* hs_all, float* gates, float* hn,
                    void* seq_ws, int T, int R, int H, void* stream);
int dtc_gru_seq_fwd_pair(const float* const* gi, const float* const* h0, const float* const* W_hh, const float* const* b_hh,
                         float* const* hs_all, float* const* gates, float* const* hn, void* const* seq_ws, int T, int R, int H, void* stream);

/* Several recurrences of ONE shape (T, R, H) advanced together -- the actor's and the critic's `Memory` of ActorCriticRecurrent
 * (actor_critic_recurrent.py:45-46: memory_a, memory_c; both are evaluated on the same mini-batch of padded trajectories,
 * ppo.py:265-272): with count == 2 every time step is ONE launch for both (a time step of one recurrence is a latency-bound launch of
 * ~200-300 workgroups; two of them on two streams overlap by only ~20 %).  Results are bit-identical to dtc_gru_fwd / dtc_gru_bwd on
 * each item.  Settings or shapes without the split-path step kernels, and count == 1: the single calls, one after the other.
 * dtc_gru_bwd_multi never forms the W_hh weight gradients: dgh_all of item i sits at its workspace + dtc_gru_dgh_offset(T, R, H),
 * as after dtc_gru_bwd with dW_hh == NULL.  Each item brings its own workspace (>= dtc_gru_workspace bytes). */
#define DTC_GRU_MULTI_MAX 2
typedef struct DtcGruFwdItem {
    const float* gi;     /* [T,R,3H] */
    const float* h0;     /* [R,H]    */
    const float* W_hh;   /* [3H,H]   */
    const float* b_hh;   /* [3H]     */
    float* hs_all;       /* [T+1,R,H] */
    float* gates;        /* [T,R,3H] */
    float* hn;           /* [T,R,H]  */
    void* workspace;
} DtcGruFwdItem;
typedef struct DtcGruBwdItem {
    const float* dhs;    /* [T,R,H]  */
    const float* hs_all;
    const float* gates;
    const float* hn;
    const float* W_hh;
    float* dgi;          /* [T,R,3H] */
    float* dh0;          /* [R,H]    */
    void* workspace;
} DtcGruBwdItem;
int dtc_gru_fwd_multi(const DtcGruFwdItem* items, int count, int T, int R, int H, void* stream);
int dtc_gru_bwd_multi(const DtcGruBwdItem* items, int count, int T, int R, int H, void* stream);

/* ---- LSTM (torch.nn.LSTM, gate order i,f,g,o; the default `rnn_type` of actor_critic_recurrent.py:93-97) ----
 * gi [T,R,4H] = x W_ih^T + b_ih is computed by dtc_linear_fwd over all T*R rows; this runs
 *     a = gi_t + h_{t-1} W_hh^T + b_hh;  i,f,o = sig(a_i, a_f, a_o), g = tanh(a_g);  c_t = f c_{t-1} + i g;  h_t = o tanh(c_t)
 * hs_all / cs_all are [T+1,R,H]: slot 0 receives h0 / c0, slot t+1 = h_t / c_t (hs_all[1:] is nn.LSTM's output);
 * gates [T,R,4H] receives the activated (i, f, g, o).  workspace >= dtc_lstm_workspace() bytes. */
int64_t dtc_lstm_workspace(int T, int R, int H);
int dtc_lstm_fwd(const float* gi, const float* h0, const float* c0, const float* W_hh /*[4H,H]*/, const float* b_hh,
                 float* hs_all, float* cs_all, float* gates, void* workspace, int T, int R, int H, void* stream);
/* BPTT.  dhs [T,R,H] = gradient w.r.t. the outputs h_1..h_T (the final states carry no gradient).  Produces dgi [T,R,4H]
 * (gradient w.r.t. the gate pre-activations: feed it to dtc_linear_wgrad with x for W_ih / b_ih and to dtc_linear_dgrad
 * for the layer below), dW_hh [4H,H], db_hh [4H], dh0 and dc0 [R,H]. */
int dtc_lstm_bwd(const float* dhs, const float* hs_all, const float* cs_all, const float* gates, const float* W_hh,
                 float* dgi, float* dW_hh, float* db_hh, float* dh0, float* dc0, void* workspace, int T, int R, int H,
                 void* stream);

/* ---- per-kernel timing (HIP events on `stream`) used by bench.py's roofline object ---------- */
void dtc_prof_enable(int on);
/* Fills up to `cap` records; returns the number of distinct kernel classes seen. */
/* work = algorithmic work of the launches in the unit of the kernel's roofline (FLOP for the MFMA-bound GEMMs, bytes for
 * HBM-bound kernels); bytes = algorithmic HBM bytes of the GEMM launches (every operand read once, every output
 * written once; 0 where not tracked) -- the denominator of bench.py's traffic ratio. */
typedef struct DtcProfRec { char name[48]; double ms_total; double work; int64_t launches; double bytes; } DtcProfRec;
int dtc_prof_report(DtcProfRec* out, int cap);
void dtc_prof_reset(void);

/* Measurement aid: one launch of the split kernels' bare MFMA stream (24 v_mfma_f32_32x32x16_bf16 per wave and stage, three waves per
 * SIMD, register operands taken from the 64 KiB at `operands`) -- blocks * 4 * iters * 24 * 32768 bf16 FLOP.  bench.py times it on
 * zero and on random operand bits: the rate the chip SUSTAINS (it clocks to its power budget) next to the data-sheet peak. */
int dtc_probe_mfma_stream(const void* operands, int blocks, int iters, float* sink, void* stream);
int dtc_probe_mfma_stream_h2(const void* operands, int blocks, int iters, float* sink, void* stream);   /* 12 fp16 MFMAs per stage (two-term path) */
/* Debugging aid (tools/flake_probe.py, tests): `blocks` workgroups that leave `pattern` in all of their 64 KiB of LDS and in ~240 VGPRs per
 * lane, so that a following kernel's reads of LDS words / registers it never wrote depend on `pattern` (uninitialised-read detector). */
int dtc_probe_poison(uint32_t pattern, int blocks, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DTC_HIP_H */
