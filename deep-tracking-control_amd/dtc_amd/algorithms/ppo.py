"""PPO with the CE-net VAE step, on hand-written HIP kernels.

Class surface = rsl_rl/rsl_rl/algorithms/ppo.py:42-381 (`PPO.__init__` keyword list, `init_storage`,
`act`, `process_env_step`, `compute_returns`, `update` returning the 7-tuple of ppo.py:356-357,
attributes `actor_critic`, `optimizer`, `vae_optimizer`, `storage`, `learning_rate`, `transition`).

`update()` executes, per mini-batch, exactly the two optimisation steps of ppo.py:189-338
(SURVEY.md Appendix B), but as an explicit forward / backward kernel schedule:
  * no autograd graph: forward activations are kept in a reusable workspace, the backward pass is
    a fixed sequence of weight-gradient / data-gradient GEMMs with fused activation derivatives;
  * the mini-batch gather and every torch.cat are folded into the GEMM operand loaders;
  * both optimisers are one fused clip+Adam launch over a contiguous parameter range;
  * no host synchronisation inside the update: losses accumulate in a device table, the adaptive
    learning rate lives in device memory, and ONE device->host copy at the end yields the returned
    means and the new `learning_rate` (the reference syncs 7x per mini-batch);
  * under torch.distributed (one process per GPU, RCCL over xGMI) gradients of each optimiser
    range are all-reduced as one flat bucket and the KL statistic is averaged so that every rank
    takes the same learning-rate branch.
"""
from __future__ import annotations

import contextlib
import ctypes
import os

import torch

from .. import _ffi, distributed as dp, h2i, ops, tracing
from .._ffi import seg, segmat
from ..modules.actor_critic_decoder import AC_Args, ActorCriticDecoder
from ..storage import RolloutStorage

# columns of the per-step statistics table
S_RECONS, S_VEL, S_KLD, S_HEIGHT, S_VAE_GNORM, S_SURR, S_VALUE, S_ENTROPY, S_KL, S_GNORM = range(10)
STAT_COLS = 12


class FusedAdam:
    """torch.optim.Adam semantics (default betas/eps, no weight decay) over ONE contiguous range of
    the parameter arena; `state_dict()` / `load_state_dict()` use torch's Adam layout so that
    checkpoints written by the reference's `OnPolicyRunner.save` load (and vice versa)."""

    def __init__(self, arena, rng, params, lr, betas=(0.9, 0.999), eps=1e-8):
        self.arena, self.range = arena, rng
        lo, hi = rng
        dev = arena.flat.device
        self.p, self.g = arena.flat[lo:hi], arena.grad[lo:hi]
        self.exp_avg = torch.zeros(hi - lo, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(hi - lo, dtype=torch.float32, device=dev)
        self.lr_dev = torch.tensor([lr], dtype=torch.float64, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.ws = ops.workspace(_ffi.lib().dtc_adam_workspace(hi - lo), dev)
        self.step_count = 0
        self.params = list(params)
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False,
                                  maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                                  decoupled_weight_decay=False)]     # the key set of torch 2.10's Adam
        names = {id(p): k for k, p in arena_named(arena)}
        self._inside = []          # (param position, arena offset, numel) of params inside the fused range
        for i, p in enumerate(self.params):
            off, n, _ = arena.offsets[names[id(p)]]
            if lo <= off and off + n <= hi:
                self._inside.append((i, off - lo, n, tuple(p.shape)))

    def rebind(self, arena):
        """The model re-built its arena (its parameters moved to another device): point the parameter / gradient
        views at the new buffers and carry the Adam state over (the layout of an arena is a function of the model)."""
        lo, hi = self.range
        dev = arena.flat.device
        self.arena, self.p, self.g = arena, arena.flat[lo:hi], arena.grad[lo:hi]
        for name in ("exp_avg", "exp_avg_sq", "lr_dev", "gnorm", "ws"):
            setattr(self, name, getattr(self, name).to(dev))

    def set_lr(self, lr: float):
        self.lr_dev.fill_(lr)
        for g in self.param_groups:
            g['lr'] = lr

    def zero_grad(self, set_to_none=True):
        self.g.zero_()

    def step(self, max_grad_norm: float, gnorm_out=None):
        """clip_grad_norm_(range, max_grad_norm) + Adam step, one fused launch pair."""
        self.step_count += 1
        g = self.param_groups[0]
        ops.clip_adam(self.p, self.g, self.exp_avg, self.exp_avg_sq, max_grad_norm, self.lr_dev, g['betas'][0],
                      g['betas'][1], g['eps'], self.step_count, gnorm_out if gnorm_out is not None else self.gnorm,
                      self.ws)

    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, off, n, shape in self._inside:
                state[i] = dict(step=torch.tensor(float(self.step_count)),
                                exp_avg=self.exp_avg[off:off + n].view(shape).clone(),
                                exp_avg_sq=self.exp_avg_sq[off:off + n].view(shape).clone())
        groups = [{k: v for k, v in self.param_groups[0].items() if k != 'params'}]
        groups[0]['params'] = list(range(len(self.params)))
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        steps = []
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, off, n, shape in self._inside:
            st = sd['state'].get(i, sd['state'].get(str(i)))
            if st is None:
                continue
            self.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.append(int(float(st['step'])))
        self.step_count = max(steps) if steps else 0
        if sd.get('param_groups'):
            self.set_lr(float(sd['param_groups'][0]['lr']))


def arena_named(arena):
    return [(k, p) for k, p in arena._named]


class _Lanes:
    """Streams of one optimisation step: `main` (torch's current stream), `aux` = a second compute lane for the branch
    of the layer graph that is independent of the one on main, `side` = weight gradients.  `order(a, b)` is the only
    synchronisation primitive: everything launched so far on lane a happens before what lane b launches next."""

    def __init__(self, dev):
        # the second compute lane and the weight-gradient streams are high-priority HIP streams: their (short, latency-bound) kernels get
        # the workgroup slots the main lane's 768-workgroup launches free up first -- 50.3 vs 50.9 ms per step, two interleaved rounds
        # (tools/jobs/r5_prio.sh; "aux" alone 50.85, "side" alone 50.6).  DTC_LANE_PRIO=none: default priorities everywhere
        prio = os.environ.get("DTC_LANE_PRIO", "aux,side").split(",")
        # ... EXCEPT where several ranks of a job share this device (the one-GPU rehearsals of the data-parallel path over gloo; RCCL refuses
        # two ranks on one device): there the lanes are torch's pooled default-priority streams, the one configuration in which the 2-rank
        # runs never differed from run to run (0 of 30; 3 of 10 with high-priority lanes, pooled or our own; one process: 0 of 50; two
        # processes that exchange nothing: 0 of 30 -- DESIGN.md §5, profiles/r06_flake*.txt)
        shared = "DTC_LANE_PRIO" not in os.environ and dp.world_size() > 1 and dp.backend() != "nccl"
        if shared:
            prio = []
        # ... and they are streams of the library's own (dtc_stream_create), NOT entries of torch's stream pool: torch.cuda.Stream(priority=-1)
        # hands out the 32 pooled high-priority streams round-robin, and torch.distributed's gloo backend takes the work stream of every
        # collective on a device tensor from the SAME pool -- every few exchanges a collective's staging copies ran on the stream that is
        # also a compute lane here.  DTC_LANE_POOL=1: torch's pool, as until round 6
        pooled = shared or os.environ.get("DTC_LANE_POOL", "0") == "1"
        self._own = []

        def make(name):
            if pooled:
                return torch.cuda.Stream(device=dev, **(dict(priority=-1) if name in prio else {}))
            h = ctypes.c_void_p()
            with torch.cuda.device(dev):
                _ffi.check(_ffi.lib().dtc_stream_create(int(name in prio), ctypes.byref(h)), "dtc_stream_create")
            self._own.append(h)
            return torch.cuda.ExternalStream(h.value, device=dev)
        self.side = make("side")
        self.side2 = make("side")          # image chain: the narrow layers' (latency-bound) weight-gradient group beside the wide one
        self.aux = make("aux")
        self.main = None                    # torch's current stream at the start of the step
        self.two_lanes = False
        self._events, self._ev_next = [], 0
        self.joined, self.joined2 = torch.cuda.Event(), torch.cuda.Event()
        self.side_busy = self.side2_busy = False

    def __del__(self):
        try:
            for h in self._own:
                _ffi.lib().dtc_stream_destroy(h)        # (hipStreamDestroy returns at once; the stream goes when its work has drained)
            self._own = []
        except Exception:                                # interpreter shutdown: the library / torch may be gone already
            pass

    def begin(self, two_lanes):
        self.main = torch.cuda.current_stream()
        self.two_lanes = two_lanes
        if two_lanes:
            self.order("main", "aux")

    def _stream(self, which):
        return self.aux if which == "aux" else self.main

    def lane(self, which):
        """Context: launches inside go to the named lane ("main" | "aux"); a no-op with one lane."""
        if self.two_lanes and which == "aux":
            return torch.cuda.stream(self.aux)
        return contextlib.nullcontext()

    def order(self, src, dst):
        if not self.two_lanes or src == dst:
            return
        ev = self.event()
        ev.record(self._stream(src))
        self._stream(dst).wait_event(ev)

    def event(self):
        if self._ev_next == len(self._events):
            self._events.append(torch.cuda.Event())
        ev = self._events[self._ev_next]
        self._ev_next += 1
        return ev

    def join(self):
        """Main waits for the second lane and for every weight gradient of this optimiser step."""
        self.order("aux", "main")
        if self.side_busy:
            self.joined.record(self.side)
            self.main.wait_event(self.joined)
            self.side_busy = False
        if self.side2_busy:
            self.joined2.record(self.side2)
            self.main.wait_event(self.joined2)
            self.side2_busy = False
        self._ev_next = 0


class _TrainWorkspace(_Lanes):
    """Activation / gradient buffers of one mini-batch step (allocated once per batch size).

    Every layer's input-gradient gets its OWN buffer (`g(name, width)`): the weight gradients run on a side
    stream concurrently with the data-gradient chain (see PPO._bwd), so a buffer that a later layer's dgrad
    would overwrite may still be being read.  26 buffers x B x <=693 floats ~ 1 GB at B = 24576."""

    def __init__(self, B, dev, num_actions):
        super().__init__(dev)
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.B, self._dev = B, dev
        # VAE-only forward activations
        self.c1, self.c2, self.rec = e(B, 64), e(B, 128), e(B, 53)
        self.d1, self.d2, self.hr = e(B, 512), e(B, 512), e(B, 693)
        # loss gradients
        self.g_rec, self.g_hr = e(B, 53), e(B, 693)
        self.dlt = e(B, 512)
        self.dmulv, self.dz = e(B, 35), e(B, 16)
        self.dmean, self.dval = e(B, num_actions), e(B, 1)
        self._g = {}
        lib = _ffi.lib()
        shapes = [(512, 693), (512, 512), (693, 512), (512, 584), (512, 752), (256, 512), (128, 256), (64, 531),
                  (128, 64), (53, 128), (128, 265), (64, 128), (35, 64), (num_actions, 128), (1, 128)]
        self.wg = ops.workspace(max(lib.dtc_linear_wgrad_workspace(B, n, k) for n, k in shapes), dev)
        self.loss_ws = ops.workspace(lib.dtc_loss_workspace(B), dev)
        self.hpart = torch.zeros(int(lib.dtc_linear_fwd_mse_parts(B, 693)), dtype=torch.float64, device=dev)
        self.gws = self.gws2 = self.gws_img = None
        self.pending, self.held = [], []      # queued weight-gradient jobs; operands of flushed jobs (alive until the join)
        self.pending_img = []                 # queued weight-gradient jobs whose operands are activation images
        self._imgs, self.live_img = {}, set()
        self.narrow_wgrad = False             # the step in flight runs the image chain: its fp32 weight-gradient jobs are the narrow layers

    def img(self, name, width=None):
        """Operand image (h2i.HImage) of the [B, width] activation / gradient `name`, allocated on first use."""
        im = self._imgs.get(name)
        if im is None:
            im = self._imgs[name] = h2i.HImage(self.B, width, self._dev)
        return im

    def value(self, name):
        """fp32 view for tests: decoded from the image when the last step kept `name` as an image only."""
        return self._imgs[name].to_tensor() if name in self.live_img else getattr(self, name)

    def group_ws_img(self, jobs):
        need = h2i.wgrad_group_workspace_bytes(jobs, self.B)
        if self.gws_img is None or self.gws_img.numel() * self.gws_img.element_size() < need:
            torch.cuda.synchronize()
            self.gws_img = ops.workspace(need, self._dev)
        return self.gws_img

    MAX_GROUP = 12           # jobs per grouped weight-gradient launch (MAX_JOBS of csrc/wgrad.hip)

    def group_ws(self, jobs, split=None, lane2=False):
        """Partial-slab workspace of a grouped weight-gradient launch, grown on demand.  `lane2`: the launch goes to the second
        weight-gradient stream -- its own slab, so that launches on `side` and `side2` of one step never share partials."""
        need = ops.wgrad_group_workspace_bytes(jobs, self.B, split)
        name = "gws2" if lane2 else "gws"
        cur = getattr(self, name, None)
        if cur is None or cur.numel() * cur.element_size() < need:
            torch.cuda.synchronize()            # nothing may still be reading the buffer being replaced
            cur = ops.workspace(need, self._dev)
            setattr(self, name, cur)
        return cur

    def wgrad_ws(self, N, K):
        """Split-partials workspace, grown on demand (rare: first use of a larger layer shape)."""
        need = ops.wgrad_workspace_bytes(self.B, N, K)
        if self.wg.numel() * self.wg.element_size() < need:
            torch.cuda.synchronize()            # nothing may still be reading the buffer being replaced
            self.wg = ops.workspace(need, self._dev)
        return self.wg

    def g(self, name, width):
        t = self._g.get(name)
        if t is None:
            t = self._g[name] = torch.empty(self.B, width, dtype=torch.float32, device=self._dev)
        return t


class PPO:
    actor_critic: ActorCriticDecoder

    def __init__(self, actor_critic, num_learning_epochs=5, num_mini_batches=4, clip_param=0.2, gamma=0.99, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.01, learning_rate=5.e-4, max_grad_norm=1.0,
                 use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device='cpu'):
        self.device = device
        self.desired_kl = desired_kl
        self.schedule = schedule
        self.learning_rate = learning_rate
        self.actor_critic = actor_critic
        self.actor_critic.to(self.device)
        self.storage = None
        if not hasattr(actor_critic, "vae"):
            # same failure the reference has (ppo.py:79) -- only ActorCriticDecoder-like models train here
            raise AttributeError(f"'{type(actor_critic).__name__}' object has no attribute 'vae'")
        self.optimizer = None
        self.vae_optimizer = None
        if torch.device(device).type == "cuda":
            self._build_optimizers()
        self.transition = RolloutStorage.Transition()
        self.clip_param = clip_param
        self.num_learning_epochs = num_learning_epochs
        self.num_mini_batches = num_mini_batches
        self.value_loss_coef = value_loss_coef
        self.entropy_coef = entropy_coef
        self.gamma = gamma
        self.lam = lam
        self.max_grad_norm = max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.num_adaptation_module_substeps = 1
        self._tws = {}
        # weight gradients on a side stream, concurrent with the data-gradient chain (DTC_OVERLAP_WGRAD=0: serial)
        self.overlap_wgrad = os.environ.get("DTC_OVERLAP_WGRAD", "1") != "0"
        self.fuse_heads = True       # dtc_ppo_heads_loss in the policy step
        # two compute lanes (needs the side stream: both lanes' weight gradients share one partials workspace)
        self.overlap_lanes = os.environ.get("DTC_OVERLAP_LANES", "1") != "0"
        # weight gradients queued per gradient bucket and run as one grouped launch (group_wgrad = False: per layer)
        self.group_wgrad = True
        # weight images of the split-path layers: built by ONE launch at the start of each optimisation step (ops.WeightImages)
        self.group_wimages = ops.WIMG
        self._wimages = {}
        # rollout step (policy sample + value + log-prob) replayed from a HIP graph: OFF by default -- measured slower
        # than the eager launches on ROCm 7.2 (793 k vs 829 k env-steps/s end to end, tools/soak.py); graph_rollout = True
        self.graph_rollout = False
        # data parallel: per-bucket gradient all-reduce on the weight-gradient stream, overlapping the rest of the
        # backward pass (DTC_DP_OVERLAP=0: one all-reduce per optimiser step after the join)
        self.overlap_exchange = os.environ.get("DTC_DP_OVERLAP", "1") != "0"
        # terrain-decoder output layer fused with the height loss (fuse_height_loss = False: separate layer + loss kernel)
        self.fuse_height_loss = True
        # ReLU layers record their output signs in the forward epilogue; the data gradient reads 1 bit instead of the saved
        # 4-byte activation (relu_masks = False: derivative through the saved activations; bit-identical results)
        self.relu_masks = True
        self.pack_inputs = True
        # the wide stacks on operand images (dtc_amd/h2i.py: activations / gradients live in HBM as the fp16 (hi, lo) planes the GEMM
        # kernels read by LDS-DMA, per-row exponents, written once by the producing epilogue); DTC_H2I=0: round 4's converting kernels
        self.use_images = os.environ.get("DTC_H2I", "1") != "0"
        self.side2_wgrad = True
        # ... and the narrow CE-net stacks (128 / 64 / 35 / 53 columns) on them as well: same kernels (a column tile partly used), their
        # weight gradients as extra jobs of the wide layers' grouped launches instead of single-pass groups of their own (narrow_images = False)
        self.narrow_images = True
        # ... and as chains: the CE-net encoder / decoder stacks (every layer <= 128 columns wide) as ONE launch per direction
        # (h2i.linear_fwd_chain / linear_dgrad_chain: the workgroup of a row tile runs layer after layer; DTC_H2I_CHAIN=0: a launch per layer)
        self.narrow_chains = os.environ.get("DTC_H2I_CHAIN", "1") != "0"
        self.tail_chains = False           # the actor's / critic's tails as chains too: built and measured slower (round 6)
        self._wsets = {}                   # phase -> h2i.WeightSet (weight images, one grouped launch per phase)
        # tests: callable(fw, which) run between the forward and the backward pass of a step ("vae" | "ppo"); the parity tests
        # use it to teacher-force the ReLU sign records (fw.relu_mask buffers) so that fp32 knife edges -- pre-activations that
        # are 0 within rounding and land on different sides in two correct implementations -- do not enter the gradient comparison
        self.after_forward_hook = None
        self._rollout_graphs = {}
        self.capture_grads, self.captured = False, {}      # tests: snapshot of the (pre-clip) gradient arena
        self.last_update_stats = None      # [steps, STAT_COLS] table of the last update (host tensor)

    def _build_optimizers(self):
        ac = self.actor_critic
        arena = ac.ensure_arena()
        arena._named = list(ac.named_parameters())
        self.optimizer = FusedAdam(arena, arena.main_range, ac.parameters(), lr=self.learning_rate)   # ppo.py:78
        self.vae_optimizer = FusedAdam(arena, arena.vae_range, ac.vae.parameters(), lr=5.e-4)          # ppo.py:79
        # data parallel: every rank starts from rank 0's weights whatever its local seed was (the averaged gradient
        # then keeps them identical); a no-op on one rank
        dp.broadcast_parameters_(arena.flat)

    def _require_gpu(self):
        if self.optimizer is None:
            raise _ffi.DtcError("dtc_amd.PPO computes on an MI355X only (device='cuda:N'); there is no CPU fallback")

    def _arena(self):
        """The model's current arena; optimisers that still hold views of a replaced arena are re-bound."""
        arena = self.actor_critic.ensure_arena()
        for opt in (self.optimizer, self.vae_optimizer):
            if opt is not None and opt.arena is not arena:
                arena._named = list(self.actor_critic.named_parameters())
                opt.rebind(arena)
                # weight images are keyed by the weights' addresses: those of the replaced arena are dead (and would be rebuilt every phase)
                self._wsets.clear()
        return arena

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape,
                     obs_history_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape,
                                      obs_history_shape, action_shape, self.device)

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    # ---------------------------------------------------------------- rollout side (ppo.py:137-172)
    def _policy_step(self, obs, privileged_obs, obs_history, base_vel, rew_buf=None):
        """The kernels of one rollout step (ppo.py:137-155): policy sample, value, log-prob."""
        ac = self.actor_critic
        actions = ac.act(obs, obs_history, privileged_obs, rew_buf)        # no autograd graph exists: nothing to detach
        values = ac.evaluate(obs, privileged_obs, base_vel)
        logp = ac.get_actions_log_prob(actions)                            # the log-prob dtc_gaussian_act already produced
        return actions, values, logp, ac.action_mean.detach(), ac.action_std.detach()

    def _policy_step_graphed(self, obs, privileged_obs, obs_history, base_vel):
        """Same kernels replayed from a HIP graph (about 25 launch-bound kernels on [N, .] rows per env step):
        captured once per batch size after a warm-up call; inputs are copied into the capture's static buffers."""
        key = (obs.shape[0], obs.device)
        g = self._rollout_graphs.get(key)
        if g is None:
            ins = [torch.empty_like(t, memory_format=torch.contiguous_format).float() for t in (obs, privileged_obs, obs_history, base_vel)]
            for dst, src in zip(ins, (obs, privileged_obs, obs_history, base_vel)):
                dst.copy_(src)
            s = torch.cuda.Stream(device=obs.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                               # warm-up: workspaces, lazy initialisation
                self._policy_step(*ins)
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = self._policy_step(*ins)
            g = self._rollout_graphs[key] = (graph, ins, outs)
        graph, ins, outs = g
        for dst, src in zip(ins, (obs, privileged_obs, obs_history, base_vel)):
            dst.copy_(src)
        graph.replay()
        return outs

    def act(self, obs, privileged_obs, obs_history, base_vel, rew_buf=None):
        self._require_gpu()
        tr = self.transition
        if self.graph_rollout and type(self).act is PPO.act and not self.actor_critic.is_recurrent:
            out = self._policy_step_graphed(obs, privileged_obs, obs_history, base_vel)
        else:
            out = self._policy_step(obs, privileged_obs, obs_history, base_vel, rew_buf)
        tr.actions, tr.values, tr.actions_log_prob, tr.action_mean, tr.action_sigma = out
        tr.observations = obs
        tr.critic_observations = obs
        tr.privileged_observations = privileged_obs
        tr.observation_histories = obs_history
        tr.base_vel = base_vel
        return tr.actions

    def process_env_step(self, rewards, dones, next_obs, infos):
        tr = self.transition
        tr.rewards = rewards
        tr.dones = dones
        tr.next_observations = next_obs
        # bootstrapping on time outs (ppo.py:162-163) happens inside the fused store: r + gamma * V * time_out
        time_outs = infos['time_outs'] if 'time_outs' in infos else None
        self.storage.add_transitions(tr, time_outs=time_outs, gamma=self.gamma)
        tr.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs, last_critic_privileged_obs, last_base_vel):
        self._require_gpu()
        last_values = self.actor_critic.evaluate(last_critic_obs, last_critic_privileged_obs, last_base_vel).detach()
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    # ---------------------------------------------------------------- update (ppo.py:174-357)
    def _train_ws(self, B):
        dev = self.actor_critic.std.device
        ws = self._tws.get((B, dev))
        if ws is None:
            ws = self._tws[(B, dev)] = _TrainWorkspace(B, dev, self.actor_critic.num_actions)
        return ws

    @staticmethod
    def _world():
        return dp.world_size()

    def _allreduce_grads(self, opt):
        """One flat all-reduce per optimiser step (no-op on a single rank): the fallback when the per-bucket exchange did
        not run.  The main optimiser's range starts at offset 0, so its exchange carries the KL header too."""
        if self._world() == 1:
            return
        arena, (lo, hi) = self.actor_critic.arena, opt.range
        dp.allreduce_mean_(arena.grad_full[0:arena.HEADER + hi] if lo == 0 else opt.g)

    def _exchange_bucket(self, tw, name):
        """Weight gradients of the bucket: flushed here.  Data parallel: average gradient bucket `name` of the arena over
        the ranks as soon as its last weight gradient has been queued.  All weight gradients of this trainer run on the
        side stream, so the all-reduce is issued THERE: it is ordered after them (and, through the lane events below,
        after everything the compute lanes have written so far: the GRU gradients of the recurrent trainers, the KL
        header) and overlaps the data-gradient chain still running on the compute lanes -- the decoders' / heads' bucket
        travels over xGMI while the encoders run backward.  Returns True when it took place."""
        self._flush_wgrads(tw)
        if self._world() == 1 or not (self.overlap_exchange and self.overlap_wgrad):
            return False
        for lane in ((tw.main, tw.aux) if tw.two_lanes else (torch.cuda.current_stream(),)):
            ev = tw.event()
            ev.record(lane)
            tw.side.wait_event(ev)
        if tw.side2_busy:                                          # the narrow layers' group ran beside the wide one: the bucket needs both
            ev = tw.event()
            ev.record(tw.side2)
            tw.side.wait_event(ev)
        with torch.cuda.stream(tw.side):
            dp.allreduce_mean_(self.actor_critic.arena.exchange_view(name))
        tw.side_busy = True
        return True

    def _kl_to_header(self, stats):
        """Data parallel, adaptive schedule: the local KL mean goes into slot 0 of the gradient header and is averaged with the first
        gradient bucket of the policy step (no collective of its own on the critical path).  The loss's finalize launch deposits it
        there itself (DtcPpoCfg.kl_mirror, see _loss_cfg): no copy between the loss and the backward pass.  (Until round 6 this was a
        4-byte device-to-device torch copy_ on the compute stream, right behind the loss kernels: DESIGN.md §5, the run-to-run
        differences of the 2-rank tests.)"""
        return

    def _lr_from_header(self, stats):
        """After the exchange: the averaged KL drives the learning-rate rule (identical on every rank) and replaces the
        local value in the statistics table (written by the same launch)."""
        if self._world() > 1 and self._adaptive():
            ops.lr_adapt(self.actor_critic.arena.kl_slot, self.optimizer.lr_dev, float(self.desired_kl), kl_out=stats[S_KL:S_KL + 1])

    def _image_mode(self, fw):
        """This step runs the wide stacks on operand images (see ActorCriticDecoder.images_ok)."""
        return (self.use_images and self.relu_masks and self.group_wgrad and self.fuse_height_loss and self.actor_critic.images_ok(fw)
                and fw.relu_mask("t1", 512) is not None)

    def _wset(self, phase):
        ws = self._wsets.get(phase)
        if ws is None:
            ws = self._wsets[phase] = h2i.WeightSet()
        return ws

    def _bwd_img(self, tw, L, dZimg, Ximg, wcol0=0, bias=True):
        """The weight gradient of a layer (or of the column block of it that meets operand image Ximg) whose operands are images:
        queued for the bucket's image-operand grouped launch."""
        tw.pending_img.append((dZimg, Ximg, L.gW, wcol0, L.gb if bias else None))
        if len(tw.pending_img) == tw.MAX_GROUP:
            self._flush_wgrads(tw)

    def _bwd(self, tw, L, dZ, X, dX=None, Xsaved=None, act_prev=None, mask=None, split=None):
        """Backward of one dense layer on fp32 operands.  The weight gradient (dW = dZ^T X) is off the critical path -- only the
        optimiser step (and the data-parallel exchange) needs it -- so it is QUEUED: `_flush_wgrads` runs all queued
        layers of a gradient bucket as one grouped launch on the side stream, where it overlaps with the data-gradient
        chain of the layers below.  (group_wgrad = False: one launch pair per layer, issued right here.)  `split=False`: the
        single-pass fp32 kernels whatever the layer's shape (the narrow layers beside the operand-image chain)."""
        if self.group_wgrad:
            tw.pending.append((dZ, X, L.gW, L.gb))
            if len(tw.pending) == tw.MAX_GROUP:
                self._flush_wgrads(tw)
        elif self.overlap_wgrad:
            ev = tw.event()
            ev.record()                          # dZ and X are final on the main stream here
            tw.side.wait_event(ev)
            ops.linear_wgrad(dZ, X, L.gW, L.gb, tw.wgrad_ws(L.n_out, L.n_in), M=tw.B, stream_ptr=tw.side.cuda_stream)
            tw.side_busy = True
        else:
            ops.linear_wgrad(dZ, X, L.gW, L.gb, tw.wgrad_ws(L.n_out, L.n_in), M=tw.B)
        if dX is not None:
            ops.linear_dgrad(dZ, L.W, dX, Xsaved, act_prev, M=tw.B, mask=mask, split=split)

    def _flush_wgrads(self, tw):
        """Launch the queued weight gradients (everything both compute lanes have issued so far is their input): one grouped
        launch for the layers whose operands are images, one for the (narrow) layers with fp32 operands."""
        if not tw.pending and not tw.pending_img:
            return
        jobs, tw.pending = tw.pending, []
        jobs_img, tw.pending_img = tw.pending_img, []
        narrow = tw.narrow_wgrad                               # image chain: the fp32 jobs are the narrow layers -> single-pass kernels
        two = self.overlap_wgrad and bool(jobs_img) and bool(jobs) and narrow and self.side2_wgrad   # the narrow group runs BESIDE the wide one
        ws = tw.group_ws(jobs, False if narrow else None, lane2=two) if jobs else None
        ws_img = tw.group_ws_img(jobs_img) if jobs_img else None
        sp = sp2 = None
        if self.overlap_wgrad:
            lanes = (tw.main, tw.aux) if tw.two_lanes else (torch.cuda.current_stream(),)
            for lane in lanes:
                ev = tw.event()
                ev.record(lane)
                tw.side.wait_event(ev)
                if two:
                    tw.side2.wait_event(ev)
            sp = sp2 = tw.side.cuda_stream
            tw.side_busy = True
            if two:
                sp2 = tw.side2.cuda_stream
                tw.side2_busy = True
        if jobs_img:
            tw.held.append(h2i.wgrad_group(jobs_img, tw.B, ws_img, stream_ptr=sp))
        if jobs:
            tw.held.append(ops.wgrad_group(jobs, tw.B, ws, stream_ptr=sp2, split=False if narrow else None))

    def _join(self, tw):
        self._flush_wgrads(tw)
        tw.narrow_wgrad = False
        tw.join()
        tw.held.clear()

    def _terrain_encoder_backward(self, fw, tw, flat, idx, wset=None):
        L = self.actor_critic.L
        rm = self.relu_masks
        if "dlt" in tw.live_img and {"t1", "t2"} <= fw.live_img:
            # image chain: d l_t arrived as an image (the decoders' / the actor's data gradient wrote it), t1 / t2 and the packed
            # heights are images; nothing of this chain exists as fp32
            g_te2i, g_te1i = tw.img("g_te2", L["te2"].n_in), tw.img("g_te1", L["te1"].n_in)
            self._bwd_img(tw, L["te2"], tw.img("dlt"), fw.img("t2"))
            h2i.linear_dgrad(tw.img("dlt"), L["te2"].W, None, g_te2i, mask=fw.relu_mask("t2", 512, rm), wset=wset)
            self._bwd_img(tw, L["te1"], g_te2i, fw.img("t1"))
            h2i.linear_dgrad(g_te2i, L["te1"].W, None, g_te1i, mask=fw.relu_mask("t1", 512, rm), wset=wset)
            self._bwd_img(tw, L["te0"], g_te1i, fw.cur["p_te"])
            tw.live_img |= {"g_te2", "g_te1"}
            return
        g_te2, g_te1 = tw.g("te2", 512), tw.g("te1", 512)
        self._bwd(tw, L["te2"], tw.dlt, fw.t2, g_te2, fw.t2, "relu", fw.relu_mask("t2", 512, rm))
        self._bwd(tw, L["te1"], g_te2, fw.t1, g_te1, fw.t1, "relu", fw.relu_mask("t1", 512, rm))
        self._bwd(tw, L["te0"], g_te1, segmat([seg(flat["privileged_observations"], 0, 693, gather=True)], idx))

    def _cenet_encoder_backward(self, fw, tw, flat, idx, split=None, wset=None):
        L = self.actor_critic.L
        if wset is not None and {"e1", "e"} <= fw.live_img:
            # image chain: d mulv (written as fp32 by the latent kernel, 35 wide) is packed once; from there every gradient of the
            # encoder exists as an image only, and its three weight gradients join the bucket's image-operand grouped launch
            dmi = tw.img("dmulv", L["head"].n_out).pack(tw.dmulv)
            g_headi, g_ce1i = tw.img("g_head", L["head"].n_in), tw.img("g_ce1", L["ce1"].n_in)
            self._bwd_img(tw, L["head"], dmi, fw.img("e"))
            self._bwd_img(tw, L["ce1"], g_headi, fw.img("e1"))
            chain = [dict(dZimg=dmi, W=L["head"].W, dXimg=g_headi),
                     dict(dZimg=g_headi, W=L["ce1"].W, dXimg=g_ce1i, mask=fw.relu_mask("e1", L["ce0"].n_out, self.relu_masks))]
            if self.narrow_chains:
                h2i.linear_dgrad_chain(chain, wset=wset)
            else:
                for c in chain:
                    h2i.linear_dgrad(c["dZimg"], c["W"], None, c["dXimg"], mask=c.get("mask"), wset=wset)
            self._bwd_img(tw, L["ce0"], g_ce1i, fw.cur["p_hist"])
            tw.live_img |= {"g_head", "g_ce1"}
            return
        g_head, g_ce1 = tw.g("head", 64), tw.g("ce1", 128)
        self._bwd(tw, L["head"], tw.dmulv, fw.e, g_head, None, None, split=split)
        self._bwd(tw, L["ce1"], g_head, fw.e1, g_ce1, fw.e1, "relu", fw.relu_mask("e1", 128, self.relu_masks), split=split)
        self._bwd(tw, L["ce0"], g_ce1, segmat([seg(flat["observation_histories"], 0, flat["observation_histories"].shape[1],
                                                   gather=True)], idx))

    def _vae_step(self, fw, tw, flat, idx, eps, stats):
        """ppo.py:197-254: CE-net / terrain auto-encoder losses, backward, clip, Adam(5e-4).

        Two compute lanes: the CE-net branch (encoder -> latent -> decoder; small layers) runs on `aux` next to
        the terrain auto-encoder (512-wide layers) on the main stream; `tw.order(a, b)` marks each point where
        one branch needs the other's result."""
        ac = self.actor_critic
        L = ac.L
        _ffi.lib().dtc_set_concurrency_hint(int(bool(self.overlap_wgrad)))     # weight gradients on the side stream
        early = self._run_phase("vae", fw, tw, self._vae_forward_backward, flat, idx, eps, stats)
        if not early:
            self._allreduce_grads(self.vae_optimizer)
        if self.capture_grads:
            self.captured["vae"] = ac.arena.grad.clone()
        self.vae_optimizer.step(self.max_grad_norm, stats[S_VAE_GNORM:S_VAE_GNORM + 1])

    def _images(self, phase):
        """ops.WeightImages block of an optimisation step (group_wimages = False: every call builds its own image)."""
        if not self.group_wimages:
            return contextlib.nullcontext()
        if phase not in self._wimages:
            self._wimages[phase] = ops.WeightImages()
        return self._wimages[phase]

    def _run_phase(self, phase, fw, tw, body, *args):
        """Forward + backward of one optimisation step.  Image chain: the phase's weight images are rebuilt by one grouped launch in
        front of the lanes' fork (the optimiser wrote the weights since they were last built); nothing else is kept on the host.
        Otherwise: round 4's converting kernels inside their ops.WeightImages block."""
        tw.narrow_wgrad = self._image_mode(fw)
        if tw.narrow_wgrad:
            wset = self._wset(phase)
            wset.rebuild()
            return body(fw, tw, *args, wset)
        with self._images(phase):
            return body(fw, tw, *args, None)

    def _vae_forward_backward(self, fw, tw, flat, idx, eps, stats, wset=None):
        ac = self.actor_critic
        L = ac.L
        tw.begin(self.overlap_lanes and self.overlap_wgrad)
        dec_in = segmat([seg(fw.z, 0, 16), seg(fw.mulv, 0, 3), seg(fw.lt, 0, 512)])
        rm = self.relu_masks
        im = wset is not None                                      # operand-image chain for the 512-wide stacks
        ns = False if im else None                                 # ... beside it the narrow layers run on the single-pass fp32 kernels
        tw.live_img.clear()
        imn = im and self.narrow_images                            # ... unless they run on images too (their weight gradients then
        with tw.lane("aux"):                                       # ride in the wide layers' grouped launches)
            ac.cenet_forward_(fw, flat["observation_histories"], eps, idx, masks=rm, split=ns, images=imn, wset=wset)
        ac.terrain_encoder_(fw, flat["privileged_observations"], idx, masks=rm, images=im, wset=wset, lt_fp32=not imn)
        tw.order("main", "aux")                                    # l_t feeds the CE-net decoder
        with tw.lane("aux"):
            if imn:
                # decoder input = the [z | mu[:, :3]] image the latent kernel wrote (19 wide) beside the l_t image; c1 / c2 leave as images only
                p_d = fw.cur["p_zmu"]
                c1i, c2i = tw.img("c1", L["cd0"].n_out), tw.img("c2", L["cd1"].n_out)
                chain = [dict(X=[p_d, fw.img("lt")], W=L["cd0"].W, b=L["cd0"].b, Yimg=c1i, act="relu", mask=fw.relu_mask("c1", 64, rm)),
                         dict(X=c1i, W=L["cd1"].W, b=L["cd1"].b, Yimg=c2i, act="relu", mask=fw.relu_mask("c2", 128, rm)),
                         dict(X=c2i, W=L["cd2"].W, b=L["cd2"].b, Y=tw.rec)]
                if self.narrow_chains:
                    h2i.linear_fwd_chain(chain, wset=wset)
                else:
                    for c in chain:
                        h2i.linear_fwd(c["X"], c["W"], c["b"], c.get("Y"), c.get("Yimg"), c.get("act"), mask=c.get("mask"), wset=wset)
                tw.live_img |= {"c1", "c2"}
            else:
                ops.linear_fwd(dec_in, L["cd0"].W, L["cd0"].b, tw.c1, "relu", M=tw.B, mask=fw.relu_mask("c1", 64, rm), split=ns)
                ops.linear_fwd(tw.c1, L["cd1"].W, L["cd1"].b, tw.c2, "relu", mask=fw.relu_mask("c2", 128, rm), split=ns)
                ops.linear_fwd(tw.c2, L["cd2"].W, L["cd2"].b, tw.rec, None, split=ns)
        if im:                                                     # terrain decoder on images: d1 / d2 never exist as fp32
            d1i, d2i = tw.img("d1", L["td0"].n_out), tw.img("d2", L["td1"].n_out)
            h2i.linear_fwd(fw.img("lt"), L["td0"].W, L["td0"].b, None, d1i, "relu", mask=fw.relu_mask("d1", 512, rm), wset=wset)
            h2i.linear_fwd(d1i, L["td1"].W, L["td1"].b, None, d2i, "relu", mask=fw.relu_mask("d2", 512, rm), wset=wset)
            tw.live_img |= {"d1", "d2"}
        else:
            ops.linear_fwd(fw.lt, L["td0"].W, L["td0"].b, tw.d1, "relu", mask=fw.relu_mask("d1", 512, rm))
            ops.linear_fwd(tw.d1, L["td1"].W, L["td1"].b, tw.d2, "relu", mask=fw.relu_mask("d2", 512, rm))
        if self.after_forward_hook is not None:
            self.after_forward_hook(fw, "vae")
        # The loss kernel joins the two branches, but only the CE-net decoder's backward needs its output (dL/d recons,
        # the direct part of d mulv): the terrain decoder's dL/dY comes out of its own output layer.  The loss therefore
        # runs on `aux`; the main lane goes straight from the terrain decoder's forward into its backward.
        if self.fuse_height_loss:
            # output layer of the terrain decoder + its MSE against priv[..., 696:] in one kernel: dL/d height_recon comes
            # out of the GEMM epilogue, height_recon itself never reaches HBM
            if im:                                                 # dL/d height_recon leaves as an image only (its two consumers read images)
                n_hp = h2i.linear_fwd_mse(tw.img("d2"), L["td2"].W, L["td2"].b, flat["privileged_observations"], 696, idx, None,
                                          tw.img("g_hr", L["td2"].n_out), tw.hpart, wset=wset)
                tw.live_img |= {"g_hr"}
            else:
                n_hp = ops.linear_fwd_mse(tw.d2, L["td2"].W, L["td2"].b, flat["privileged_observations"], 696, idx, tw.g_hr, tw.hpart)
            tw.order("main", "aux")
            with tw.lane("aux"):
                ops.vae_loss_fused(tw.rec, fw.mulv, flat["next_observations"], flat["base_vel"], idx, tw.g_rec, tw.dmulv,
                                   tw.hpart, n_hp, stats[S_RECONS:S_RECONS + 4], tw.loss_ws,
                                   drec_img=tw.img("g_rec", L["cd2"].n_out) if imn else None)
        else:
            ops.linear_fwd(tw.d2, L["td2"].W, L["td2"].b, tw.hr, None)
            tw.order("main", "aux")
            with tw.lane("aux"):
                ops.vae_loss(tw.rec, tw.hr, fw.mulv, flat["next_observations"], flat["privileged_observations"],
                             flat["base_vel"], idx, tw.g_rec, tw.g_hr, tw.dmulv, stats[S_RECONS:S_RECONS + 4], tw.loss_ws)
            tw.order("aux", "main")
        # CE-net decoder (aux, short): its input gradient fans out to z, mu[:, :3] (accumulating onto the loss's direct
        # part) and l_t (plain write) -- it finishes long before the terrain decoder's chain on main reaches its last
        # layer, whose input gradient is then ADDED to d l_t (a + b == b + a: same bits as the reverse order)
        g_cd2, g_cd1 = tw.g("cd2", 128), tw.g("cd1", 64)
        dst = segmat([seg(tw.dz, 0, 16), seg(tw.dmulv, 0, 3, accumulate=True), seg(tw.dlt, 0, 512)])
        with tw.lane("aux"):
            if imn:
                g_reci = tw.img("g_rec", L["cd2"].n_out)              # written by the loss kernel
                g_cd2i, g_cd1i = tw.img("g_cd2", L["cd2"].n_in), tw.img("g_cd1", L["cd1"].n_in)
                self._bwd_img(tw, L["cd2"], g_reci, tw.img("c2"))
                self._bwd_img(tw, L["cd1"], g_cd2i, tw.img("c1"))
                chain = [dict(dZimg=g_reci, W=L["cd2"].W, dXimg=g_cd2i, mask=fw.relu_mask("c2", 128, rm)),
                         dict(dZimg=g_cd2i, W=L["cd1"].W, dXimg=g_cd1i, mask=fw.relu_mask("c1", 64, rm))]
                if self.narrow_chains:
                    h2i.linear_dgrad_chain(chain, wset=wset)
                else:
                    for c in chain:
                        h2i.linear_dgrad(c["dZimg"], c["W"], None, c["dXimg"], mask=c.get("mask"), wset=wset)
                self._bwd_img(tw, L["cd0"], g_cd1i, fw.img("lt"), wcol0=19)          # columns of dW that meet l_t ...
                self._bwd_img(tw, L["cd0"], g_cd1i, fw.cur["p_zmu"], wcol0=0, bias=False)      # ... and [z | mu[:, :3]]
                # W's columns [19, 531) first (d l_t: four whole 128-column tiles, 16-byte stores), then [0, 19) (dz | d mu[:, :3] accumulating)
                h2i.linear_dgrad(g_cd1i, L["cd0"].W, segmat([seg(tw.dlt, 0, 512), seg(tw.dz, 0, 16), seg(tw.dmulv, 0, 3, accumulate=True)]), None,
                                 window=[(19, 512), (0, 19)], wset=wset)
                tw.live_img |= {"g_cd2", "g_cd1"}
            else:
                self._bwd(tw, L["cd2"], tw.g_rec, tw.c2, g_cd2, tw.c2, "relu", fw.relu_mask("c2", 128, rm), split=ns)
                self._bwd(tw, L["cd1"], g_cd2, tw.c1, g_cd1, tw.c1, "relu", fw.relu_mask("c1", 64, rm), split=ns)
                self._bwd(tw, L["cd0"], g_cd1, dec_in, dst, None, None, split=ns)
        # terrain decoder (main)
        if im:
            g_td2i, g_td1i = tw.img("g_td2", L["td2"].n_in), tw.img("g_td1", L["td1"].n_in)
            self._bwd_img(tw, L["td2"], tw.img("g_hr"), tw.img("d2"))
            h2i.linear_dgrad(tw.img("g_hr"), L["td2"].W, None, g_td2i, mask=fw.relu_mask("d2", 512, rm), wset=wset)
            self._bwd_img(tw, L["td1"], g_td2i, tw.img("d1"))
            h2i.linear_dgrad(g_td2i, L["td1"].W, None, g_td1i, mask=fw.relu_mask("d1", 512, rm), wset=wset)
            tw.order("aux", "main")                                # d l_t of the CE-net decoder is written first (fp32) ...
            # ... and added to this layer's product; the sum leaves as the image the terrain encoder's backward reads
            self._bwd_img(tw, L["td0"], g_td1i, fw.img("lt"))
            h2i.linear_dgrad(g_td1i, L["td0"].W, None, tw.img("dlt", L["td0"].n_in), add=tw.dlt, wset=wset)
            tw.live_img |= {"g_td2", "g_td1", "dlt"}
        else:
            g_td2, g_td1 = tw.g("td2", 512), tw.g("td1", 512)
            self._bwd(tw, L["td2"], tw.g_hr, tw.d2, g_td2, tw.d2, "relu", fw.relu_mask("d2", 512, rm))
            self._bwd(tw, L["td1"], g_td2, tw.d1, g_td1, tw.d1, "relu", fw.relu_mask("d1", 512, rm))
            tw.order("aux", "main")                                # d l_t of the CE-net decoder is written first
            self._bwd(tw, L["td0"], g_td1, fw.lt, segmat([seg(tw.dlt, 0, 512, accumulate=True)]), None, None)
        early = self._exchange_bucket(tw, "vae_only")              # decoder gradients are complete (queued on `side`)
        self._terrain_encoder_backward(fw, tw, flat, idx, wset)
        with tw.lane("aux"):
            ops.cenet_latent_bwd(tw.dmulv, tw.dz, eps, fw.mulv, fw.mask, fw.info, fw.lat_ws)
            self._cenet_encoder_backward(fw, tw, flat, idx, split=ns, wset=wset if imn else None)
        if early:
            self._exchange_bucket(tw, "shared")
        self._join(tw)
        return early

    def _ppo_step(self, fw, tw, flat, idx, eps, stats, cfg):
        """ppo.py:265-335: policy / value forward with the freshly updated VAE, PPO losses, backward,
        clip, Adam(adaptive lr).  Lanes: CE-net encoder + critic on `aux`, terrain encoder + actor on main."""
        ac = self.actor_critic
        L = ac.L
        act = AC_Args.activation
        _ffi.lib().dtc_set_concurrency_hint(int(bool(self.overlap_wgrad)))     # weight gradients on the side stream
        early = self._run_phase("ppo", fw, tw, self._ppo_forward_backward, flat, idx, eps, stats, cfg)
        if not early:
            self._allreduce_grads(self.optimizer)
        self._lr_from_header(stats)
        if self.capture_grads:
            self.captured["main"] = ac.arena.grad.clone()
        self.optimizer.step(self.max_grad_norm, stats[S_GNORM:S_GNORM + 1])

    def _ppo_forward_backward(self, fw, tw, flat, idx, eps, stats, cfg, wset=None):
        ac = self.actor_critic
        L = ac.L
        act = AC_Args.activation
        tw.begin(self.overlap_lanes and self.overlap_wgrad)
        im = wset is not None                                      # operand-image chain for the wide stacks
        ns = False if im else None
        tw.live_img.clear()
        obs, priv = flat["observations"], flat["privileged_observations"]
        imn = im and self.narrow_images
        with tw.lane("aux"):
            ac.cenet_forward_(fw, flat["observation_histories"], eps, idx, masks=self.relu_masks, split=ns, images=imn, wset=wset)
        ac.terrain_encoder_(fw, priv, idx, masks=self.relu_masks, images=im, wset=wset, lt_fp32=not im)
        tw.order("aux", "main")                                    # z, mu feed the actor
        # output layers + losses + their data gradients in one launch when the last hidden width allows it (fuse_heads)
        # (its partial-sum workspace holds 4096 blocks of 64 rows: larger mini-batches take the unfused kernels)
        fuse = self.fuse_heads and fw.a3.shape[1] == fw.v3.shape[1] and fw.a3.shape[1] in (64, 128, 256) and tw.B <= 4096 * 64
        a_cols = None
        if im:
            # layer-0 inputs as operand images: the critic's whole input (gathered rollout rows) is packed once; the actor's is the
            # l_t image the terrain encoder just wrote + the packed narrow block [obs | z | mu[:, :3]] (W's columns 72.. and 0..71)
            with tw.lane("aux"):
                Xc = ac.packed_input(fw, "p_c", ac.critic_input(obs, flat["base_vel"], priv, idx), idx, reuse=True)
            if imn:        # ... as two images: the gathered observations (packed once per update and mini-batch) and the latent kernel's [z | mu[:, :3]]
                Xa = [fw.img("lt"), ac.packed_input(fw, "p_obs", segmat([seg(obs, 0, ac.num_obs, gather=True)], idx), idx, reuse=True), fw.cur["p_zmu"]]
                a_cols = [ac.num_obs + 19, 0, ac.num_obs]
            else:
                Xa = [fw.img("lt"), ac.packed_input(fw, "p_a", segmat([seg(obs, 0, ac.num_obs, gather=True), seg(fw.z, 0, 16), seg(fw.mulv, 0, 3)], idx))]
                a_cols = [ac.num_obs + 19, 0]
        elif self.pack_inputs:                                     # the narrow leading blocks of both layer-0 inputs packed into dense operands
            with tw.lane("aux"):
                Xc = ac.critic_input_packed(obs, flat["base_vel"], priv, idx, tw.g("pack_c", ac.num_obs + 3), tw.B)
            Xa = ac.actor_input_packed(fw, obs, idx, tw.g("pack_a", ac.num_obs + 19))
        else:
            Xc = ac.critic_input(obs, flat["base_vel"], priv, idx)
            Xa = ac.actor_input(fw, obs, idx)
        himg = None                                                # fused heads in the image chain: the kernel writes its four gradients as images too
        if imn and fuse and fw.a3.shape[1] <= 128:
            himg = (tw.img("G_a3", fw.a3.shape[1]), tw.img("G_c3", fw.v3.shape[1]), tw.img("dmean", tw.dmean.shape[1]), tw.img("dval", 1))
        with tw.lane("aux"):
            ac.critic_forward_(fw, obs, flat["base_vel"], priv, idx, head=not fuse, X=Xc, images=im, wset=wset, last_img=himg is not None)
        ac.actor_forward_(fw, obs, idx, head=not fuse, X=Xa, images=im, wset=wset, cols=a_cols, last_img=himg is not None)
        tw.order("aux", "main")
        if self.after_forward_hook is not None:
            self.after_forward_hook(fw, "ppo")
        g_c3, g_a3 = tw.g("c3", 128), tw.g("a3", 128)
        if fuse:
            ops.ppo_heads_loss(fw.a3, fw.v3, L["a3"].W, L["a3"].b, L["c3"].W, L["c3"].b, act, ac.std_view, flat["actions"],
                               flat["actions_log_prob"], flat["mu"], flat["sigma"], flat["advantages"], flat["returns"],
                               flat["values"], idx, cfg, fw.mean, fw.val, tw.dmean, tw.dval,
                               None if himg is not None else g_a3, None if himg is not None else g_c3,     # (images only: no fp32 copy)
                               ac.std_grad, stats[S_SURR:S_SURR + 4], self.optimizer.lr_dev, tw.loss_ws, imgs=himg)
        else:
            ops.ppo_loss(fw.mean, ac.std_view, fw.val, flat["actions"], flat["actions_log_prob"], flat["mu"],
                         flat["sigma"], flat["advantages"], flat["returns"], flat["values"], idx, cfg, tw.dmean, tw.dval,
                         ac.std_grad, stats[S_SURR:S_SURR + 4], self.optimizer.lr_dev, tw.loss_ws)
        self._kl_to_header(stats)
        tw.order("main", "aux")
        tw.dmulv.zero_()
        if im:
            self._ppo_backward_images(fw, tw, Xc, Xa, a_cols, g_a3, g_c3, fuse, wset, himg)
        else:
            # critic (aux)
            g_c2, g_c1 = tw.g("c2", 256), tw.g("c1", 512)
            with tw.lane("aux"):
                self._bwd(tw, L["c3"], tw.dval, fw.v3, None if fuse else g_c3, fw.v3, act)      # fused: weight gradient only
                self._bwd(tw, L["c2"], g_c3, fw.v2, g_c2, fw.v2, act)
                self._bwd(tw, L["c1"], g_c2, fw.v1, g_c1, fw.v1, act)
                self._bwd(tw, L["c0"], g_c1, Xc)
            # actor (main); layer-0 input gradient fans out to z, mu[:, :3], l_t (observations need none)
            g_a2, g_a1 = tw.g("a2", 256), tw.g("a1", 512)
            self._bwd(tw, L["a3"], tw.dmean, fw.a3, None if fuse else g_a3, fw.a3, act)
            self._bwd(tw, L["a2"], g_a3, fw.a2, g_a2, fw.a2, act)
            self._bwd(tw, L["a1"], g_a2, fw.a1, g_a1, fw.a1, act)
            dst = segmat([seg(None, 0, ac.num_obs), seg(tw.dz, 0, 16), seg(tw.dmulv, 0, 3), seg(tw.dlt, 0, 512)])
            self._bwd(tw, L["a0"], g_a1, Xa, dst, None, None)
        early = self._exchange_bucket(tw, "main_only")             # actor + critic + std gradients are complete
        tw.order("main", "aux")                                    # dz, d mu[:, :3] (and d l_t) are written
        self._terrain_encoder_backward(fw, tw, flat, idx, wset)    # needs d l_t only: starts right away on main
        with tw.lane("aux"):
            ops.cenet_latent_bwd(tw.dmulv, tw.dz, eps, fw.mulv, fw.mask, fw.info, fw.lat_ws)
            self._cenet_encoder_backward(fw, tw, flat, idx, split=ns, wset=wset if imn else None)
        if early:
            self._exchange_bucket(tw, "shared")
        self._join(tw)
        return early

    def _ppo_backward_images(self, fw, tw, Xc, Xa, a_cols, g_a3, g_c3, fuse, wset, himg=None):
        """Backward of the actor / critic bodies on operand images.  The heads' gradients (fp32, 128 wide) are packed into images; from
        there every gradient of the two bodies exists as an image only.  The ELU derivative reads the fp32 copy of the saved
        activation the forward pass kept next to its image."""
        ac = self.actor_critic
        L = ac.L
        act = AC_Args.activation
        n_a, n_c = L["a2"].n_out, L["c2"].n_out
        with tw.lane("aux"):                                       # critic
            if himg is not None:                                   # the heads kernel wrote G_c3 / d value as images: the output layer's
                G_c3 = himg[1]                                     # weight gradient joins the bucket's image-operand launch, nothing to pack
                self._bwd_img(tw, L["c3"], himg[3], fw.img("v3"))
            else:
                self._bwd(tw, L["c3"], tw.dval, fw.v3, None if fuse else g_c3, fw.v3, act, split=False)      # fused: weight gradient only
                G_c3 = tw.img("G_c3", n_c).pack(g_c3)
            g_c2i, g_c1i = tw.img("g_c2", L["c2"].n_in), tw.img("g_c1", L["c1"].n_in)
            self._bwd_img(tw, L["c2"], G_c3, fw.img("v2"))
            self._bwd_img(tw, L["c1"], g_c2i, fw.img("v1"))
            self._tail_dgrad(G_c3, L["c2"], g_c2i, fw.v2, L["c1"], g_c1i, fw.v1, act, wset)
            self._bwd_img(tw, L["c0"], g_c1i, Xc)
        # actor (main); layer-0 input gradient fans out to z, mu[:, :3] (fp32) and l_t (image); the observations need none
        if himg is not None:
            G_a3 = himg[0]
            self._bwd_img(tw, L["a3"], himg[2], fw.img("a3"))
        else:
            self._bwd(tw, L["a3"], tw.dmean, fw.a3, None if fuse else g_a3, fw.a3, act, split=False)
            G_a3 = tw.img("G_a3", n_a).pack(g_a3)
        g_a2i, g_a1i = tw.img("g_a2", L["a2"].n_in), tw.img("g_a1", L["a1"].n_in)
        self._bwd_img(tw, L["a2"], G_a3, fw.img("a2"))
        self._bwd_img(tw, L["a1"], g_a2i, fw.img("a1"))
        self._tail_dgrad(G_a3, L["a2"], g_a2i, fw.a2, L["a1"], g_a1i, fw.a1, act, wset)
        nb = ac.num_obs + 19                                       # width of the narrow block [obs | z | mu[:, :3]]
        for i, (xi, c0) in enumerate(zip(Xa, a_cols)):             # columns of dW that meet l_t, then the narrow block(s)
            self._bwd_img(tw, L["a0"], g_a1i, xi, wcol0=c0, bias=i == 0)
        # ONE launch over W's columns [72, 584) then [53, 72): d l_t as an image (four 128-column tiles), dz | d mu[:, :3] as fp32 (a fifth)
        h2i.linear_dgrad(g_a1i, L["a0"].W, segmat([seg(None, 0, 512), seg(tw.dz, 0, 16), seg(tw.dmulv, 0, 3)]), tw.img("dlt", 512),
                         window=[(nb, 512), (ac.num_obs, 19)], wset=wset)
        tw.live_img |= {"dlt", "g_a2", "g_a1", "g_c2", "g_c1"}

    def _tail_dgrad(self, G3, L2, g2i, x2, L1, g1i, x1, act, wset):
        """Data gradients of a body's tail 128 -> 256 -> 512 (actor_critic_decoder.py:323-349 backward): ONE launch -- the workgroup of a
        row tile runs the two column tiles of the 256-wide gradient, then the four of the 512-wide one (round 6; bit-identical) -- measured
        slower than a launch per layer (see modules/actor_critic_decoder.py: TAIL_CHAINS), so `tail_chains` is False."""
        chain = [dict(dZimg=G3, W=L2.W, dXimg=g2i, Xsaved=x2, act=act), dict(dZimg=g2i, W=L1.W, dXimg=g1i, Xsaved=x1, act=act)]
        if self.tail_chains:
            h2i.linear_dgrad_chain(chain, wset=wset)
        else:
            for c in chain:
                h2i.linear_dgrad(c["dZimg"], c["W"], None, c["dXimg"], Xsaved=c["Xsaved"], act=c["act"], wset=wset)

    def _adaptive(self):
        return self.desired_kl is not None and self.schedule == 'adaptive'

    def step_minibatch(self, idx, eps1, eps2, which="both", stats=None):
        """One mini-batch of `update` on the stored rollout: the VAE step, the PPO step, or both
        (`which` in {"vae", "ppo", "both"}); returns the statistics row (host tensor, columns S_*) and the
        learning rate after adaptation.  Used by teacher-forced parity tests."""
        self._require_gpu()
        st, ac = self.storage, self.actor_critic
        self._arena()
        dev = ac.std.device
        idx = idx.to(dev).contiguous()
        B = idx.numel()
        flat = {k: st.flat(k) for k in self._FLAT_NAMES}
        fw, tw = ac._fwd_ws(B), self._train_ws(B)
        self._amax_static(flat, fw)
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1
        fw.pack_gen, fw.pack_slot = self._pack_gen, 0
        self.optimizer.set_lr(self.learning_rate)
        stats = torch.zeros(STAT_COLS, dtype=torch.float32, device=dev) if stats is None else stats.to(dev)
        try:
            if which in ("vae", "both"):
                self._vae_step(fw, tw, flat, idx, eps1.to(dev).contiguous(), stats)
            if which in ("ppo", "both"):
                self._ppo_step(fw, tw, flat, idx, eps2.to(dev).contiguous(), stats, self._loss_cfg())
        finally:
            ops.amax_static_clear()
            fw.pack_gen = None
        self.learning_rate = float(self.optimizer.lr_dev.item())
        for g in self.optimizer.param_groups:
            g['lr'] = self.learning_rate
        return stats.cpu(), self.learning_rate

    _FLAT_NAMES = ("observations", "next_observations", "privileged_observations", "observation_histories", "actions",
                   "values", "advantages", "returns", "actions_log_prob", "mu", "sigma", "base_vel")

    _AMAX_STATIC = ("observations", "next_observations", "privileged_observations", "observation_histories", "base_vel")

    def _amax_static(self, flat, fw=None):
        """Two-term fp16 GEMM path (ops.H2) on round 4's converting kernels: the rollout tensors that enter GEMMs as gathered operands bring
        the amax of the whole stored tensor -- they do not change during the update, so it is computed once here, before the compute lanes
        fork.  The operand-image chain keeps no amax records at all: where the whole step runs on it (`fw` given and _image_mode(fw), with
        the narrow layers on images too) the five passes over up to 546 MB are skipped (round 6: 0.33 ms per update)."""
        if fw is not None and type(self)._ppo_step is PPO._ppo_step and self.narrow_images and self._image_mode(fw):
            ops.amax_static_clear()
            return
        if ops.SPLIT and ops.H2:
            ops.amax_static_clear()
            for k in self._AMAX_STATIC:
                if k in flat and flat[k].dtype == torch.float32 and flat[k].dim() == 2:
                    ops.amax_static(flat[k])

    def _loss_cfg(self):
        cfg = _ffi.DtcPpoCfg()
        cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef = self.clip_param, self.value_loss_coef, self.entropy_coef
        cfg.desired_kl = float(self.desired_kl) if self.desired_kl is not None else 0.0
        cfg.use_clipped_value_loss = int(bool(self.use_clipped_value_loss))
        cfg.adaptive_schedule = int(self._adaptive() and self._world() == 1)
        # data parallel: the finalize launch also deposits the KL mean in the gradient header (slot 0), which the first bucket's
        # all-reduce averages over the ranks
        cfg.kl_mirror = self.actor_critic.arena.kl_slot.data_ptr() if (self._adaptive() and self._world() > 1) else None
        return cfg

    def update(self, perm=None, eps1=None, eps2=None, return_stats=False):
        """One PPO update over the stored rollout.  `perm` / `eps1` / `eps2` optionally inject the random
        draws the reference takes from torch's generator (rollout_storage.py:165, actor_critic_decoder.py:283)
        so that parity tests can feed both implementations the same numbers."""
        self._require_gpu()
        st, ac = self.storage, self.actor_critic
        self._arena()
        dev = ac.std.device
        nmb, epochs = self.num_mini_batches, self.num_learning_epochs
        B = (st.num_envs * st.num_transitions_per_env) // nmb
        steps = nmb * epochs
        if perm is None or eps1 is None or eps2 is None:
            seed = ops.draw_seed()             # counter-based device draws (csrc/rng.hip): one launch each, no sort
            perm = ops.randperm(nmb * B, dev, seed) if perm is None else perm
            eps1 = ops.randn((steps, B, 16), dev, seed + 1) if eps1 is None else eps1
            eps2 = ops.randn((steps, B, 16), dev, seed + 2) if eps2 is None else eps2
        perm = perm.to(dev).contiguous()
        flat = {k: st.flat(k) for k in self._FLAT_NAMES}
        fw, tw = ac._fwd_ws(B), self._train_ws(B)
        self._amax_static(flat, fw)
        cfg = self._loss_cfg()
        self.optimizer.set_lr(self.learning_rate)
        stats = torch.zeros(steps, STAT_COLS, dtype=torch.float32, device=dev)
        lr_hist = torch.zeros(steps, dtype=torch.float64, device=dev) if return_stats else None
        k = 0
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1
        fw.pack_gen = self._pack_gen                   # the packed rollout rows of a mini-batch serve all five epochs (packed_input)
        try:
            for _ in range(epochs):
                for i in range(nmb):
                    fw.pack_slot = i
                    idx = perm[i * B:(i + 1) * B]
                    with tracing.span("vae_step"):
                        self._vae_step(fw, tw, flat, idx, eps1[k], stats[k])
                    with tracing.span("ppo_step"):
                        self._ppo_step(fw, tw, flat, idx, eps2[k], stats[k], cfg)
                    if lr_hist is not None:
                        lr_hist[k:k + 1].copy_(self.optimizer.lr_dev)
                    k += 1
        finally:                                       # (an exception mid-update must not leave the generation live: a later direct step would
            fw.pack_gen = None                         # find the packed rows of THIS rollout under a recycled index address)
            ops.amax_static_clear()                    # the storage is about to be refilled: its amax slots are void
        # the single device -> host synchronisation of the update
        host = stats.cpu()
        self.learning_rate = float(self.optimizer.lr_dev.item())
        for g in self.optimizer.param_groups:
            g['lr'] = self.learning_rate
        self.last_update_stats = host
        m = host.double().mean(dim=0)
        st.clear()
        out = (float(m[S_VALUE]), float(m[S_SURR]), 0.0, 0, float(m[S_RECONS]), float(m[S_VEL]), float(m[S_KLD]))
        if return_stats:
            return out, host, lr_hist.cpu()
        return out
