"""ActorCriticDecoder on hand-written HIP kernels.

Same constructor, attribute names, `state_dict()` keys and initialisation order as the
reference (rsl_rl/rsl_rl/modules/actor_critic_decoder.py:91-302 `Vae`, :305-451, :540-551
`ActorCriticDecoder`), so reference checkpoints load and `torch.manual_seed(s)` gives the same
initial weights.  Differences in *how* it computes:

  * all parameters live in ONE flat fp32 device buffer (`ParamArena`) ordered so that each of
    PPO's two optimisers (ppo.py:78-79) owns one contiguous range -> one fused clip+Adam launch,
    one RCCL all-reduce per optimiser step; `nn.Parameter.data` are views into it;
  * `latent_mu` and `latent_var` are adjacent in the arena and run as one [35,64] head GEMM;
  * forward passes (`act`, `evaluate`, `act_inference`) call the C ABI (no autograd graph);
    the training step (manual backward) is driven by `dtc_amd.algorithms.ppo.PPO`.
There is no CPU path: calling a compute method without a GPU + libdtc_hip.so raises.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

import os

from .. import _ffi, h2i, ops
from .._ffi import seg, segmat

# The actor's / critic's tails 512 -> 256 -> 128 as one launch per direction (round 6): built, bit-identical, and measured
# SLOWER -- 50.2 vs ~49.0 ms per step on one box, two interleaved rounds (860 instead of 940 launches): a row tile's workgroup then runs 3
# (forward) or 6 (backward) column tiles one after the other, 384 workgroups instead of 768 + 384, and the tails sit on the compute lanes
# beside the weight-gradient stream, where the narrower launch is the worse neighbour.  Off; the kernels keep the capability (tested).
TAIL_CHAINS = False
NARROW_CHAINS = os.environ.get("DTC_H2I_CHAIN", "1") != "0"     # the CE-net encoder's three layers as one launch (see cenet_forward_)


class AC_Args:
    """Dimensions of the reference's AC_Args (actor_critic_decoder.py:11-88) that the hot path uses."""
    init_noise_std = 1.0
    actor_hidden_dims = [512, 256, 128]
    critic_hidden_dims = [512, 256, 128]
    activation = 'elu'
    terrain_latent = 512
    terrain_encoder_branch_input_dims = [693]
    terrain_encoder_branch_latent_dims = [terrain_latent]
    terrain_encoder_branch_hidden_dims = [[512, 512]]
    terrain_decoder_branch_input_dims = [terrain_latent]
    terrain_decoder_branch_output_dims = [693]
    terrain_decoder_branch_hidden_dims = [[512, 512]]
    cenet_encoder_branch_input_dims = [53 * 5]
    cenet_encoder_branch_latent_dims = [64]
    cenet_encoder_branch_hidden_dims = [[128]]
    cenet_decoder_branch_input_dims = [19 + terrain_latent]
    cenet_decoder_branch_output_dims = [53]
    cenet_decoder_branch_hidden_dims = [[64, 128]]
    gb_encoder_input_dims = [128]
    gb_encoder_hidden_dims = [[128]]
    gb_encoder_latent_dims = [64]
    memory_mlp_input_dims = [53 * 5 + terrain_latent]
    memory_mlp_hidden_dims = [[256, 128]]
    memory_mlp_latent_dims = [terrain_latent]
    rnn_type = 'gru'
    rnn_num_layers = 2
    rnn_hidden_size = 50


def _layer_init(layer, std=np.sqrt(2), bias_const=0.0):
    torch.nn.init.orthogonal_(layer.weight, std)
    torch.nn.init.constant_(layer.bias, bias_const)
    return layer


def _branch(in_dim, hidden, out_dim, act):
    """Linear(default init) -> act -> [Linear(orthogonal 0.01) -> act]* -> Linear(orthogonal 0.01)."""
    dims = [in_dim] + list(hidden) + [out_dim]
    mods = [nn.Linear(dims[0], dims[1]), act]
    for i in range(1, len(dims) - 1):
        mods.append(_layer_init(nn.Linear(dims[i], dims[i + 1]), 0.01))
        if i < len(dims) - 2:
            mods.append(act)
    return nn.Sequential(*mods)


def get_activation(act_name):
    table = dict(elu=nn.ELU, selu=nn.SELU, relu=nn.ReLU, crelu=nn.ReLU, lrelu=nn.LeakyReLU, tanh=nn.Tanh,
                 sigmoid=nn.Sigmoid)
    if act_name not in table:
        print("invalid activation function!")
        return None
    return table[act_name]()


class Dense:
    """One nn.Linear as seen by the kernels: views of weight/bias/grads inside the arena."""
    __slots__ = ("W", "b", "gW", "gb", "act", "n_out", "n_in")

    def __init__(self, W, b, gW, gb, act):
        self.W, self.b, self.gW, self.gb, self.act = W, b, gW, gb, act
        self.n_out, self.n_in = W.shape


class ParamArena:
    """Flat parameter / gradient buffers.  Layout (element offsets):
        [ main-only: actor_body, critic_body, std (or std, actor, critic, memory_a, memory_c) | shared: cenet_encoder, (latent_mu|latent_var),
          terrain_encoder | vae-only: cenet_decoder, terrain_decoder | unused: memory_mlp, gb_encoder ]
    main optimiser range = [0, end(shared));  VAE optimiser range = [start(shared), end(vae-only)).
    The parameters of `unused` never receive a gradient in PPO.update."""

    HEADER = 4

    def exchange_view(self, name):
        """The tensor the data-parallel exchange of gradient bucket `name` all-reduces (bucket 0 includes the header)."""
        lo, hi = self.buckets[name]
        return self.grad_full[0:self.HEADER + hi] if lo == 0 else self.grad[lo:hi]

    @property
    def kl_slot(self):
        return self.grad_full[0:1]

    def __init__(self, model: "ActorCriticDecoder"):
        named = dict(model.named_parameters())
        order, groups = [], {}

        def take(prefix):
            return [k for k in named if k.startswith(prefix)]

        # actor / critic bodies (+ the GRU memories of the recurrent composite), then std
        main_only = [k for k in named if not k.startswith("vae.") and k != "std"] + ["std"]
        heads = ["vae.latent_mu.weight", "vae.latent_var.weight", "vae.latent_mu.bias", "vae.latent_var.bias"]
        shared = take("vae.cenet_encoder.") + heads + take("vae.terrain_encoder.")
        vae_only = take("vae.cenet_decoder.") + take("vae.terrain_decoder.")
        unused = take("vae.memory_mlp.") + take("vae.gb_encoder.")
        order = main_only + shared + vae_only + unused
        assert sorted(order) == sorted(named), "parameter partition is incomplete"
        device = named["std"].device
        total = sum(named[k].numel() for k in order)
        self.flat = torch.empty(total, dtype=torch.float32, device=device)
        # gradient buffer with a 16-byte header in front: [kl, 0, 0, 0 | gradients ...].  The data-parallel exchange of the
        # first bucket (`main_only`, offset 0) sends header + bucket as ONE all-reduce, so the KL mean every rank needs for
        # the same learning-rate decision travels with the gradients (dtc_amd/distributed.py); `grad` itself starts
        # behind the header and keeps its 16-byte alignment
        self.grad_full = torch.zeros(self.HEADER + total, dtype=torch.float32, device=device)
        self.grad = self.grad_full[self.HEADER:]
        self.offsets, off = {}, 0
        for k in order:
            p = named[k]
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            self.offsets[k] = (off, n, tuple(p.shape))
            off += n
        n_main = sum(named[k].numel() for k in main_only)
        n_shared = sum(named[k].numel() for k in shared)
        n_vae = sum(named[k].numel() for k in vae_only)
        self.main_range = (0, n_main + n_shared)
        self.vae_range = (n_main, n_main + n_shared + n_vae)
        # the three gradient buckets of the data-parallel exchange, in backward-pass completion order per step
        self.buckets = dict(main_only=(0, n_main), shared=(n_main, n_main + n_shared),
                            vae_only=(n_main + n_shared, n_main + n_shared + n_vae))
        self.order = order

    def view(self, buf, name):
        off, n, shape = self.offsets[name]
        return buf[off:off + n].view(shape)

    def dense(self, wname, bname, act, rows=None):
        W, gW = self.view(self.flat, wname), self.view(self.grad, wname)
        b, gb = self.view(self.flat, bname), self.view(self.grad, bname)
        return Dense(W, b, gW, gb, act)

    def fused_head(self):
        """latent_mu (19) and latent_var (16) as one [35,64] layer (adjacent in the arena)."""
        o_w = self.offsets["vae.latent_mu.weight"][0]
        o_b = self.offsets["vae.latent_mu.bias"][0]
        assert self.offsets["vae.latent_var.weight"][0] == o_w + 19 * 64
        assert self.offsets["vae.latent_var.bias"][0] == o_b + 19
        mk = lambda buf: (buf[o_w:o_w + 35 * 64].view(35, 64), buf[o_b:o_b + 35])
        (W, b), (gW, gb) = mk(self.flat), mk(self.grad)
        return Dense(W, b, gW, gb, None)


class Vae(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        A = AC_Args
        relu = nn.ReLU()
        self.cenet_encoder = _branch(A.cenet_encoder_branch_input_dims[0], A.cenet_encoder_branch_hidden_dims[0],
                                     A.cenet_encoder_branch_latent_dims[0], relu)
        self.latent_mu = _layer_init(nn.Linear(16 * 4, 19), 0.01)
        self.latent_var = _layer_init(nn.Linear(16 * 4, 16), 0.01)
        self.cenet_decoder = _branch(A.cenet_decoder_branch_input_dims[0], A.cenet_decoder_branch_hidden_dims[0],
                                     A.cenet_decoder_branch_output_dims[0], relu)
        self.terrain_encoder = _branch(A.terrain_encoder_branch_input_dims[0], A.terrain_encoder_branch_hidden_dims[0],
                                       A.terrain_encoder_branch_latent_dims[0], relu)
        self.terrain_decoder = _branch(A.terrain_decoder_branch_input_dims[0], A.terrain_decoder_branch_hidden_dims[0],
                                       A.terrain_decoder_branch_output_dims[0], relu)
        self.memory_mlp = _branch(A.memory_mlp_input_dims[0], A.memory_mlp_hidden_dims[0],
                                  A.memory_mlp_latent_dims[0], relu)
        # the reference builds and discards a 64->128->693 stack at this point
        # (actor_critic_decoder.py:209-228); it draws from the init RNG, so the same draws are made here
        _branch(64, [128], 693, relu)
        self.gb_encoder = _branch(A.gb_encoder_input_dims[0], A.gb_encoder_hidden_dims[0],
                                  A.gb_encoder_latent_dims[0], relu)

    @staticmethod
    def layer_init(layer, std=np.sqrt(2), bias_const=0.0):
        return _layer_init(layer, std, bias_const)


class ActorCriticDecoder(nn.Module):
    is_recurrent = False

    def __init__(self, num_obs, num_critic_obs, num_actions, **kwargs):
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: "
                  + str([key for key in kwargs.keys()]))
        super().__init__()
        A = AC_Args
        act = get_activation(A.activation)
        self.num_obs, self.num_critic_obs, self.num_actions = num_obs, num_critic_obs, num_actions
        self.vae = Vae()
        self.bootstrap_threshold = 0.1
        self.actor_body = _branch(num_obs + 16 + 3 + A.terrain_encoder_branch_latent_dims[0], A.actor_hidden_dims,
                                  num_actions, act)
        self.critic_body = _branch(693 + num_obs + 3 + 15 + 12 - 24, A.critic_hidden_dims, 1, act)
        self.std = nn.Parameter(A.init_noise_std * torch.ones(num_actions))
        self.distribution = None
        self.arena: ParamArena | None = None
        self._fw = {}            # forward workspaces keyed by batch size
        self._dist = None        # (mean, sigma) of the last update_distribution

    # ------------------------------------------------------------------ arena management
    def _apply(self, fn, *a, **k):
        before = self.std.data_ptr()
        out = super()._apply(fn, *a, **k)
        if self.std.data_ptr() != before:
            # .to(other device) / .cuda() / a dtype change re-allocated the parameter storage: rebuild the arena lazily
            # (trainers notice the new arena object and re-bind their optimiser views, see FusedAdam.rebind).  A no-op
            # .to() -- e.g. OnPolicyRunner.get_inference_policy(device=same) -- leaves the parameters as arena views.
            self.arena = None
            self._fw = {}
        return out

    def ensure_arena(self) -> ParamArena:
        if self.arena is None or self.arena.flat.device != self.std.device or \
                self.std.data_ptr() != self.arena.view(self.arena.flat, "std").data_ptr():
            if not self.std.is_cuda:
                raise _ffi.DtcError("ActorCriticDecoder computes on the GPU only: move it to a HIP device "
                                    "(there is no CPU fallback)")
            self.arena = ParamArena(self)
            ar = self.arena
            self._fw = {}            # forward workspaces hold marshalled layer chains that point into the old arena
            self._build_layers(ar)
            self.std_view = ar.view(ar.flat, "std")
            self.std_grad = ar.view(ar.grad, "std")
        return self.arena

    @staticmethod
    def _vae_layers(ar):
        relu = "relu"
        d = ar.dense
        return dict(
                ce0=d("vae.cenet_encoder.0.weight", "vae.cenet_encoder.0.bias", relu),
                ce1=d("vae.cenet_encoder.2.weight", "vae.cenet_encoder.2.bias", None),
                head=ar.fused_head(),
                te0=d("vae.terrain_encoder.0.weight", "vae.terrain_encoder.0.bias", relu),
                te1=d("vae.terrain_encoder.2.weight", "vae.terrain_encoder.2.bias", relu),
                te2=d("vae.terrain_encoder.4.weight", "vae.terrain_encoder.4.bias", None),
                cd0=d("vae.cenet_decoder.0.weight", "vae.cenet_decoder.0.bias", relu),
                cd1=d("vae.cenet_decoder.2.weight", "vae.cenet_decoder.2.bias", relu),
                cd2=d("vae.cenet_decoder.4.weight", "vae.cenet_decoder.4.bias", None),
                td0=d("vae.terrain_decoder.0.weight", "vae.terrain_decoder.0.bias", relu),
                td1=d("vae.terrain_decoder.2.weight", "vae.terrain_decoder.2.bias", relu),
                td2=d("vae.terrain_decoder.4.weight", "vae.terrain_decoder.4.bias", None))

    def _build_layers(self, ar):
        elu = AC_Args.activation
        d = ar.dense
        self.L = self._vae_layers(ar)
        self.L.update(
                a0=d("actor_body.0.weight", "actor_body.0.bias", elu),
                a1=d("actor_body.2.weight", "actor_body.2.bias", elu),
                a2=d("actor_body.4.weight", "actor_body.4.bias", elu),
                a3=d("actor_body.6.weight", "actor_body.6.bias", None),
                c0=d("critic_body.0.weight", "critic_body.0.bias", elu),
                c1=d("critic_body.2.weight", "critic_body.2.bias", elu),
                c2=d("critic_body.4.weight", "critic_body.4.bias", elu),
                c3=d("critic_body.6.weight", "critic_body.6.bias", None),
            )

    # ------------------------------------------------------------------ forward workspace
    class _Fwd:
        def __init__(self, B, dev, num_actions):
            e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
            self.B = B
            self.e1, self.e, self.mulv, self.z = e(B, 128), e(B, 64), e(B, 35), e(B, 16)
            self.mask = torch.empty(B, 16, dtype=torch.uint8, device=dev)
            self.info = torch.zeros(4, dtype=torch.int32, device=dev)
            self.t1, self.t2, self.lt = e(B, 512), e(B, 512), e(B, 512)
            self.a1, self.a2, self.a3, self.mean = e(B, 512), e(B, 256), e(B, 128), e(B, num_actions)
            self.v1, self.v2, self.v3, self.val = e(B, 512), e(B, 256), e(B, 128), e(B, 1)
            self.lat_ws = torch.empty(int(_ffi.lib().dtc_cenet_workspace(B)) // 8 + 1, dtype=torch.float64, device=dev)
            self._dev, self._masks = dev, {}
            self._imgs, self.live_img = {}, set()
            self.pack_key = {}               # operand image name -> what it was packed from (see ActorCriticDecoder.packed_input)
            self.pack_gen, self.pack_slot = None, 0      # set by the trainer for the duration of an update (packed_input)
            self.cur = {}                    # logical name -> the packed image the step in flight uses

        def img(self, name, width=None):
            """Operand image (h2i.HImage) of the [B, width] activation `name`, allocated on first use."""
            im = self._imgs.get(name)
            if im is None:
                im = self._imgs[name] = h2i.HImage(self.B, width, self._dev)
            return im

        def value(self, name):
            """The activation `name` as fp32 (tests / debugging): decoded from its image when the last forward pass kept it as an
            image only (`live_img`), else the fp32 buffer."""
            return self._imgs[name].to_tensor() if name in self.live_img else getattr(self, name)

        def relu_mask(self, name, width, enabled=True):
            """Sign record of a ReLU layer's [B, width] output (training only): the data gradient reads one bit per
            element instead of the saved activation.  None when the shape is not eligible or the feature is off."""
            if not enabled or not ops.relu_mask_ok(self.B, width):
                return None
            m = self._masks.get(name)
            if m is None:
                m = self._masks[name] = ops.relu_mask(self.B, width, self._dev)
            return m

    def _fwd_ws(self, B):
        ws = self._fw.get(B)
        if ws is None:
            ws = self._fw[B] = ActorCriticDecoder._Fwd(B, self.std.device, self.num_actions)
        return ws

    def _rollout_chains(self, ws, obs, hist, priv, base_vel):
        """The forward passes of one env step as marshalled layer chains (ops.FwdChain, built once per batch size on first
        use): "ce" CE-net encoder + heads and "ta" terrain encoder + actor (when `hist` is given), "cr" critic (when
        `base_vel` is given).  Per step only the env's input tensors are re-pointed."""
        ch = getattr(ws, "chains", None)
        if ch is None:
            ch = ws.chains = {}
        L, act, B = self.L, AC_Args.activation, ws.B
        lay = lambda name, X, Y, a: (X, L[name].W, L[name].b, Y, a)
        if hist is not None:
            if "ce" not in ch:
                ch["ce"] = ops.FwdChain([lay("ce0", hist, ws.e1, "relu"), lay("ce1", ws.e1, ws.e, None),
                                         lay("head", ws.e, ws.mulv, None)], B)
                ch["ta"] = ops.FwdChain([lay("te0", segmat([seg(priv, 0, 693)]), ws.t1, "relu"), lay("te1", ws.t1, ws.t2, "relu"),
                                         lay("te2", ws.t2, ws.lt, None), lay("a0", self.actor_input(ws, obs), ws.a1, act),
                                         lay("a1", ws.a1, ws.a2, act), lay("a2", ws.a2, ws.a3, act), lay("a3", ws.a3, ws.mean, None)], B)
            ch["ce"].set_input(0, 0, hist)
            ch["ta"].set_input(0, 0, priv)
            ch["ta"].set_input(3, 0, obs)
        if base_vel is not None:
            if "cr" not in ch:
                ch["cr"] = ops.FwdChain([lay("c0", self.critic_input(obs, base_vel, priv), ws.v1, act), lay("c1", ws.v1, ws.v2, act),
                                         lay("c2", ws.v2, ws.v3, act), lay("c3", ws.v3, ws.val, None)], B)
            ch["cr"].set_input(0, 0, obs)
            ch["cr"].set_input(0, 1, base_vel)
            ch["cr"].set_input(0, 2, priv)
        return ch

    # ------------------------------------------------------------------ kernel-level forward pieces
    def cenet_forward_(self, ws, hist, eps, idx=None, masks=False, split=None, images=False, wset=None):
        """vae.cenet_forward (actor_critic_decoder.py:286-302) into ws.mulv / ws.z.  `masks`: training step -- the ReLU
        layers also record their output signs (ws.relu_mask) for the backward pass.  `split=False`: single-pass fp32 kernels.
        `images` (training step, see images_ok): the gathered history rows are packed into an operand image once per mini-batch and
        update, e1 / e leave as images only (ws.live_img) -- their consumers (next layer, weight gradient) read images --, the heads'
        output as fp32 for the latent kernel."""
        L = self.L
        X = segmat([seg(hist, 0, hist.shape[1], gather=idx is not None)], idx)
        if images and masks:
            pin = self.packed_input(ws, "p_hist", X, idx, reuse=True)
            e1i, ei = ws.img("e1", L["ce0"].n_out), ws.img("e", L["ce1"].n_out)
            chain = [dict(X=pin, W=L["ce0"].W, b=L["ce0"].b, Yimg=e1i, act="relu", mask=ws.relu_mask("e1", L["ce0"].n_out, masks)),
                     dict(X=e1i, W=L["ce1"].W, b=L["ce1"].b, Yimg=ei), dict(X=ei, W=L["head"].W, b=L["head"].b, Y=ws.mulv)]
            if NARROW_CHAINS:          # the three layers as ONE launch (h2i.linear_fwd_chain; DTC_H2I_CHAIN=0: a launch per layer)
                h2i.linear_fwd_chain(chain, wset=wset)
            else:
                for c in chain:
                    h2i.linear_fwd(c["X"], c["W"], c["b"], c.get("Y"), c.get("Yimg"), c.get("act"), mask=c.get("mask"), wset=wset)
            ws.live_img |= {"e1", "e"}
        else:
            ws.live_img -= {"e1", "e"}
            ops.linear_fwd(X, L["ce0"].W, L["ce0"].b, ws.e1, "relu", M=ws.B, mask=ws.relu_mask("e1", 128, masks), split=split)
            ops.linear_fwd(ws.e1, L["ce1"].W, L["ce1"].b, ws.e, None, split=split)
            ops.linear_fwd(ws.e, L["head"].W, L["head"].b, ws.mulv, None, split=split)
        zmu = None
        if images and masks:                    # [z | mu[:, :3]] also leaves as an operand image (the decoder's / the actor's narrow input block)
            zmu = ws.cur["p_zmu"] = ws.img("p_zmu", 19)
        ops.cenet_latent_fwd(ws.mulv, eps, ws.z, ws.mask, ws.info, ws.lat_ws, zmu_img=zmu)

    def images_ok(self, ws):
        """The operand-image chain (dtc_amd/h2i.py: the wide stacks' activations live in HBM as the fp16 (hi, lo) planes the GEMM
        kernels read by LDS-DMA, written once by the producing epilogue) runs on training steps whose mini-batch is a whole number of
        128-row tiles (ReLU sign records, exponent blocks of the weight gradients)."""
        return ops.SPLIT and ws.B % 128 == 0

    def packed_input(self, ws, name, X, idx=None, reuse=False):
        """Operand image `name` of the fp32 operand X (DtcSegMat: the gathered rollout rows, narrow hand-over tensors).
        `reuse`: X is a function of the rollout storage and the mini-batch index only.  The reference's mini-batches are the SAME four
        index sets in all five epochs of an update (rollout_storage.py:165, 188-193: one randperm, then the epoch loop), so inside an
        update -- the trainer sets `ws.pack_gen` / `ws.pack_slot` for its duration -- each mini-batch's image is packed once, into its
        own buffer, and serves both optimisation steps of all epochs.  Outside an update (pack_gen None) every call packs."""
        gen = ws.pack_gen if reuse else None
        slot = ws.pack_slot if gen is not None else 0
        img = ws.img(f"{name}@{slot}", X.cols)
        key = None if gen is None else (gen, None if idx is None else (idx.data_ptr(), idx.numel()))
        if key is None or ws.pack_key.get((name, slot)) != key:
            img.pack(X, ws.B)
            ws.pack_key[(name, slot)] = key
        ws.cur[name] = img
        return img

    def terrain_encoder_(self, ws, priv, idx=None, masks=False, images=False, wset=None, lt_fp32=True, split=None):
        """`images` (training step, see images_ok): the gathered heights are packed into an operand image once per mini-batch, t1 / t2
        leave as images only (ws.live_img), l_t as an image (+ fp32 for the CE-net decoder: lt_fp32)."""
        L = self.L
        X = segmat([seg(priv, 0, 693, gather=idx is not None)], idx)
        if images and masks:
            w1, w2 = L["te0"].n_out, L["te1"].n_out
            pin = self.packed_input(ws, "p_te", X, idx, reuse=True)
            t1i, t2i, lti = ws.img("t1", w1), ws.img("t2", w2), ws.img("lt", L["te2"].n_out)
            h2i.linear_fwd(pin, L["te0"].W, L["te0"].b, None, t1i, "relu", mask=ws.relu_mask("t1", w1, masks), wset=wset)
            h2i.linear_fwd(t1i, L["te1"].W, L["te1"].b, None, t2i, "relu", mask=ws.relu_mask("t2", w2, masks), wset=wset)
            h2i.linear_fwd(t2i, L["te2"].W, L["te2"].b, ws.lt if lt_fp32 else None, lti, None, wset=wset)
            ws.live_img |= {"t1", "t2"} | (set() if lt_fp32 else {"lt"})
            return
        ws.live_img -= {"t1", "t2", "lt"}
        ops.linear_fwd(X, L["te0"].W, L["te0"].b, ws.t1, "relu", M=ws.B, mask=ws.relu_mask("t1", 512, masks), split=split)
        ops.linear_fwd(ws.t1, L["te1"].W, L["te1"].b, ws.t2, "relu", mask=ws.relu_mask("t2", 512, masks), split=split)
        ops.linear_fwd(ws.t2, L["te2"].W, L["te2"].b, ws.lt, None, split=split)

    def actor_input(self, ws, obs, idx=None):
        return segmat([seg(obs, 0, self.num_obs, gather=idx is not None), seg(ws.z, 0, 16), seg(ws.mulv, 0, 3),
                       seg(ws.lt, 0, 512)], idx)

    def critic_input(self, obs, base_vel, priv, idx=None):
        g = idx is not None
        # cat[obs, base_vel, priv[:, 693:696], priv[:, 696:]] -- the last two are adjacent columns of `priv`
        return segmat([seg(obs, 0, self.num_obs, gather=g), seg(base_vel, 0, 3, gather=g),
                       seg(priv, 693, 696, gather=g)], idx)

    def actor_input_packed(self, ws, obs, idx, buf):
        """The same operand with its three narrow blocks [obs | z | mu[:, :3]] packed into `buf` [B, 72] (one launch): two
        segments instead of four -- one 16-k stage / one column tile for the narrow part instead of three."""
        ops.pack_cols(segmat([seg(obs, 0, self.num_obs, gather=idx is not None), seg(ws.z, 0, 16), seg(ws.mulv, 0, 3)], idx), buf, ws.B)
        return segmat([seg(buf, 0, self.num_obs + 19), seg(ws.lt, 0, 512)])

    def critic_input_packed(self, obs, base_vel, priv, idx, buf, B):
        g = idx is not None
        ops.pack_cols(segmat([seg(obs, 0, self.num_obs, gather=g), seg(base_vel, 0, 3, gather=g)], idx), buf, B)
        return segmat([seg(buf, 0, self.num_obs + 3), seg(priv, 693, 696, gather=g)], idx)

    def _body_forward_(self, ws, X, names, outs, images, wset=None, cols=None, last_img=False):
        """First three layers of an actor / critic body.  `images`: X is an operand image (or a list of them, `cols` = the first
        column of the layer's weight each one meets); the two hidden activations leave as fp32 (the ELU derivative of the backward pass
        reads it) AND as images, the layers after the first read their input by LDS-DMA; the last one writes fp32 only."""
        L, act = self.L, AC_Args.activation
        l0, l1, l2 = (L[n] for n in names)
        o0, o1, o2 = (getattr(ws, n) for n in outs)
        if images:
            i0, i1 = ws.img(outs[0], l0.n_out), ws.img(outs[1], l1.n_out)
            h2i.linear_fwd(X, l0.W, l0.b, o0, i0, act, wset=wset, cols=cols)
            # (last_img: the third activation as an image as well -- the X operand of the output layer's image-operand weight gradient)
            i2 = ws.img(outs[2], l2.n_out) if last_img else None
            if TAIL_CHAINS:        # the tail 512 -> 256 -> 128 as ONE launch: a row tile's workgroup runs both column tiles of the 256-wide
                                   # layer, then the 128-wide one (round 6; bit for bit the two launches)
                h2i.linear_fwd_chain([dict(X=i0, W=l1.W, b=l1.b, Y=o1, Yimg=i1, act=act), dict(X=i1, W=l2.W, b=l2.b, Y=o2, Yimg=i2, act=act)],
                                     wset=wset)
            else:
                h2i.linear_fwd(i0, l1.W, l1.b, o1, i1, act, wset=wset)
                h2i.linear_fwd(i1, l2.W, l2.b, o2, i2, act, wset=wset)
            return
        ops.linear_fwd(X, l0.W, l0.b, o0, act, M=ws.B)
        ops.linear_fwd(o0, l1.W, l1.b, o1, act)
        ops.linear_fwd(o1, l2.W, l2.b, o2, act)

    def actor_forward_(self, ws, obs, idx=None, head=True, X=None, images=False, wset=None, cols=None, last_img=False):
        """`head=False`: stop before the output layer (the trainer's fused heads + loss kernel computes it)."""
        L = self.L
        self._body_forward_(ws, self.actor_input(ws, obs, idx) if X is None else X, ("a0", "a1", "a2"), ("a1", "a2", "a3"), images, wset, cols, last_img)
        if head:
            ops.linear_fwd(ws.a3, L["a3"].W, L["a3"].b, ws.mean, None)

    def critic_forward_(self, ws, obs, base_vel, priv, idx=None, head=True, X=None, images=False, wset=None, last_img=False):
        L = self.L
        self._body_forward_(ws, self.critic_input(obs, base_vel, priv, idx) if X is None else X, ("c0", "c1", "c2"), ("v1", "v2", "v3"), images, wset, None, last_img)
        if head:
            ops.linear_fwd(ws.v3, L["c3"].W, L["c3"].b, ws.val, None)

    # ------------------------------------------------------------------ reference API
    def reset(self, dones=None):
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self._dist[0]

    @property
    def action_std(self):
        return self._dist[1]

    @property
    def entropy(self):
        s = self._dist[1]
        return (0.5 + 0.5 * float(np.log(2 * np.pi)) + torch.log(s)).sum(dim=-1)

    def _prep(self, t):
        return t.contiguous().float()

    def update_distribution(self, observations, observations_history, privileged_obs, eps=None):
        """actor_critic_decoder.py:409-437.  `eps` ([B,16], optional) injects the reparameterisation noise."""
        self.ensure_arena()
        obs, hist, priv = self._prep(observations), self._prep(observations_history), self._prep(privileged_obs)
        B = obs.shape[0]
        ws = self._fwd_ws(B)
        if eps is None:
            eps = torch.randn(B, 16, device=obs.device)
        ch = self._rollout_chains(ws, obs, hist, priv, None)
        ch["ce"].run()
        ops.cenet_latent_fwd(ws.mulv, eps, ws.z, ws.mask, ws.info, ws.lat_ws)
        ch["ta"].run()
        self.latent_mu, self.latent_var, self.z = ws.mulv[:, :19], ws.mulv[:, 19:], ws.z
        mean = ws.mean.clone()
        self._dist = (mean, self.std_view.detach().expand_as(mean))
        self.distribution = self._dist

    def act(self, observations, observations_history, privileged_obs, rew_buf=None, eps=None, noise=None, **kwargs):
        """actor_critic_decoder.py:439-447: sample from N(mean, std)."""
        self.update_distribution(observations, observations_history, privileged_obs, eps=eps)
        mean = self._dist[0]
        B, A = mean.shape
        if noise is None:
            noise = torch.randn(B, A, device=mean.device)
        actions = torch.empty_like(mean)
        self._logp = torch.empty(B, device=mean.device)
        ops.gaussian_act(mean, self.std_view, noise, actions, self._logp)
        self._last_actions = actions
        return actions

    def get_actions_log_prob(self, actions):
        """actor_critic_decoder.py:450-451."""
        mean, sigma = self._dist
        if getattr(self, "_last_actions", None) is actions:
            return self._logp
        return (-((actions - mean) ** 2) / (2 * sigma * sigma) - torch.log(sigma)
                - float(np.log(np.sqrt(2 * np.pi)))).sum(dim=-1)

    def act_expert(self, ob):
        return self.act_teacher(ob["obs"], ob["obs_history"], ob["privileged_obs"])

    def act_teacher(self, observations, observations_history, privileged_obs):
        """actor_critic_decoder.py:504-538 (deployment path of `get_inference_policy(env_t=True)`): mean action from
        latent_mu (no sampling) and the belief b_t = m + l_t * m, m = memory_mlp(cat[hist, l_t]).  `memory_mlp` never
        receives a gradient in PPO.update, so this path runs on whatever weights the checkpoint holds."""
        ar = self.ensure_arena()
        obs, hist, priv = self._prep(observations), self._prep(observations_history), self._prep(privileged_obs)
        B, dev = obs.shape[0], obs.device
        ws, L = self._fwd_ws(B), self.L
        if "mm0" not in L:
            d = ar.dense
            L.update(mm0=d("vae.memory_mlp.0.weight", "vae.memory_mlp.0.bias", "relu"),
                     mm1=d("vae.memory_mlp.2.weight", "vae.memory_mlp.2.bias", "relu"),
                     mm2=d("vae.memory_mlp.4.weight", "vae.memory_mlp.4.bias", None))
        # deployment / rollout-side rows are independent envs: the single-pass fp32 kernels (the reference's own arithmetic) keep every
        # row to itself -- a diverged env's inf / NaN observation stays in its row (split=False; PPO.act's layer chains do the same)
        fp = dict(split=False)
        ops.linear_fwd(hist, L["ce0"].W, L["ce0"].b, ws.e1, "relu", M=B, **fp)
        ops.linear_fwd(ws.e1, L["ce1"].W, L["ce1"].b, ws.e, None, **fp)
        ops.linear_fwd(ws.e, L["head"].W, L["head"].b, ws.mulv, None, **fp)          # [:, :19] = latent_mu
        self.terrain_encoder_(ws, priv, split=False)
        m1, m2, m = (torch.empty(B, n, device=dev) for n in (L["mm0"].n_out, L["mm1"].n_out, L["mm2"].n_out))
        ops.linear_fwd(segmat([seg(hist, 0, hist.shape[1]), seg(ws.lt, 0, 512)]), L["mm0"].W, L["mm0"].b, m1, "relu", M=B, **fp)
        ops.linear_fwd(m1, L["mm1"].W, L["mm1"].b, m2, "relu", **fp)
        ops.linear_fwd(m2, L["mm2"].W, L["mm2"].b, m, None, **fp)
        b_t = m + ws.lt * m
        X = segmat([seg(obs, 0, self.num_obs), seg(ws.mulv, 3, 16), seg(ws.mulv, 0, 3), seg(b_t, 0, 512)])
        act = AC_Args.activation
        ops.linear_fwd(X, L["a0"].W, L["a0"].b, ws.a1, act, M=B, **fp)
        ops.linear_fwd(ws.a1, L["a1"].W, L["a1"].b, ws.a2, act, **fp)
        ops.linear_fwd(ws.a2, L["a2"].W, L["a2"].b, ws.a3, act, **fp)
        ops.linear_fwd(ws.a3, L["a3"].W, L["a3"].b, ws.mean, None, **fp)
        return ws.mean.clone()

    def act_student(self, observations, observations_history, privileged_obs, lidar_latent, masks=None, hidden_states=None):
        """actor_critic_decoder.py:459-502: mean action of the student head from latent_mu (no sampling) and a caller-supplied
        terrain latent.  The reference's body reads `self.cenet_encoder`, `self.latent_mu` (as a layer), `self.actor_student` and
        an exporter loaded from an absolute path, none of which its class defines; here those names mean what the rest of the class
        calls them (vae.cenet_encoder, vae.latent_mu, actor_body -- the only actor, same input width), the exporter side effect is
        not restated (its output is discarded there).  `privileged_obs`, `masks`, `hidden_states` are accepted and unused, as there."""
        self.ensure_arena()
        obs, hist, lid = self._prep(observations), self._prep(observations_history), self._prep(lidar_latent)
        B = obs.shape[0]
        if lid.shape != (B, 512):
            raise ValueError(f"act_student: lidar_latent must be ({B}, 512), got {tuple(lid.shape)}")
        ws, L = self._fwd_ws(B), self.L
        fp = dict(split=False)                          # rows are independent envs: single-pass fp32 kernels (see act_teacher)
        ops.linear_fwd(hist, L["ce0"].W, L["ce0"].b, ws.e1, "relu", M=B, **fp)
        ops.linear_fwd(ws.e1, L["ce1"].W, L["ce1"].b, ws.e, None, **fp)
        ops.linear_fwd(ws.e, L["head"].W, L["head"].b, ws.mulv, None, **fp)          # [:, :19] = latent_mu
        X = segmat([seg(obs, 0, self.num_obs), seg(ws.mulv, 3, 16), seg(ws.mulv, 0, 3), seg(lid, 0, 512)])
        act = AC_Args.activation
        ops.linear_fwd(X, L["a0"].W, L["a0"].b, ws.a1, act, M=B, **fp)
        ops.linear_fwd(ws.a1, L["a1"].W, L["a1"].b, ws.a2, act, **fp)
        ops.linear_fwd(ws.a2, L["a2"].W, L["a2"].b, ws.a3, act, **fp)
        ops.linear_fwd(ws.a3, L["a3"].W, L["a3"].b, ws.mean, None, **fp)
        return ws.mean.clone()

    def adapt_bootstrap_probability(self, rewards):
        """actor_critic_decoder.py:404-407: 1 - tanh(std / mean) of a reward buffer (unbiased std, as torch.std), a Python float.
        One fused reduction on the device (`ops.bootstrap_probability`), one host read -- the `.item()` of the reference."""
        return ops.bootstrap_probability(self._prep(rewards))

    def act_inference(self, ob):
        """Deterministic policy output (mean action) for deployment-style evaluation."""
        self.update_distribution(ob["obs"], ob["obs_history"], ob["privileged_obs"],
                                 eps=torch.zeros(ob["obs"].shape[0], 16, device=ob["obs"].device))
        return self._dist[0]

    def evaluate(self, critic_observations, privileged_observations, base_vel, **kwargs):
        """actor_critic_decoder.py:540-551."""
        self.ensure_arena()
        obs, priv, bv = self._prep(critic_observations), self._prep(privileged_observations), self._prep(base_vel)
        ws = self._fwd_ws(obs.shape[0])
        self._rollout_chains(ws, obs, None, priv, bv)["cr"].run()
        return ws.val.clone()
