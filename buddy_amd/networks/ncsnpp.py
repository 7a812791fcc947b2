"""``NCSNppTime`` -- the reference's score-network plug point (``networks.ncsnpp.NCSNppTime``,
reference ``networks/ncsnpp.py:455-506``) backed by the hand-written gfx950 network in ``libbuddy_hip.so``.

Same surface as the reference class: constructor kwargs = ``conf/network/ncsnpp.yaml`` keys (incl. nested
``stft:{n_fft,hop_length,center}``), ``nn.Module`` protocol with the reference ``state_dict`` key names
(``all_modules.N.*``, ``output_layer.*``), ``forward(x:(B,1,L) f32, time_cond:(B,) f32) -> (B,1,L)``,
differentiable w.r.t. ``x`` (the input-VJP runs in HIP; no weight gradients -- inference only).
Only the shipped architecture family is supported (biggan resblocks, input_skip/sum, output_skip, one bottleneck attention;
``fir`` False or True with the (1,3,3,1) kernel); anything else raises ``NotImplementedError`` at construction.
"""
from __future__ import annotations

import copy
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..synth import module_specs


class _Bag(nn.Module):
    """parameter container; gives nested state-dict names without any compute."""


def _variance_scaling_uniform(shape, scale, in_axis, out_axis):
    # default_init: variance_scaling(scale, 'fan_avg', 'uniform') -- reference layers.py:54-91
    scale = 1e-10 if scale == 0 else scale
    rf = np.prod(shape) / shape[in_axis] / shape[out_axis]
    fan_in, fan_out = shape[in_axis] * rf, shape[out_axis] * rf
    var = scale / ((fan_in + fan_out) / 2)
    return (torch.rand(*shape) * 2.0 - 1.0) * np.sqrt(3 * var)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cnoise, scal, net):
        B, L = x.shape
        lib = _lib.require_gpu()
        x = x.contiguous()
        y = torch.empty_like(x)
        save = 1 if ctx.needs_input_grad[0] else 0
        cin = cskip = cout = None
        if scal is not None:
            cin, cskip, cout = (s.contiguous() for s in scal)
        _lib.check(lib.buddy_ncsnpp_forward(net._get_handle(), _lib.ptr(x), _lib.ptr(cnoise.contiguous()), _lib.ptr(cin),
                                            _lib.ptr(cskip), _lib.ptr(cout), _lib.ptr(y), B, L, save, _lib.stream_ptr()))
        net._fwd_id += 1
        ctx.net, ctx.fwd_id = net, net._fwd_id
        return y

    @staticmethod
    def backward(ctx, g):
        net = ctx.net
        if ctx.fwd_id != net._fwd_id:
            raise _lib.BuddyHipError("backward through a stale NCSNppTime forward: the handle keeps one tape (call "
                                     "backward before the next forward)")
        g = g.contiguous()
        gx = torch.empty_like(g)
        _lib.check(_lib.load().buddy_ncsnpp_vjp(net._get_handle(), _lib.ptr(g), _lib.ptr(gx), _lib.stream_ptr()))
        return gx, None, None, None


class NCSNppTime(nn.Module):
    ATTENTION_MODES = {"flash": 0, "bf16": 1, "f16": 2, "matrix": 3, "auto": 4}
    GEMM_MODES = {"fp32": 0, "bf16x3": 1, "f16x2": 2}

    def __init__(self, stft=None, nonlinearity="swish", nf=128, ch_mult=(1, 2, 2, 2), num_res_blocks=1,
                 attn_resolutions=(0,), resamp_with_conv=True, time_conditional=True, fir=False,
                 fir_kernel=(1, 3, 3, 1), skip_rescale=True, resblock_type="biggan", progressive="output_skip",
                 progressive_input="input_skip", progressive_combine="sum", init_scale=0.0, fourier_scale=16,
                 image_size=256, embedding_type="fourier", input_channels=2, spatial_channels=1, dropout=0.0,
                 centered=True, discriminative=False, attention=None, gemm=None, **kwargs):
        super().__init__()
        self._init_kwargs = dict(stft=stft, nonlinearity=nonlinearity, nf=nf, ch_mult=ch_mult, num_res_blocks=num_res_blocks,
                                 attn_resolutions=attn_resolutions, resamp_with_conv=resamp_with_conv, time_conditional=time_conditional,
                                 fir=fir, fir_kernel=fir_kernel, skip_rescale=skip_rescale, resblock_type=resblock_type,
                                 progressive=progressive, progressive_input=progressive_input, progressive_combine=progressive_combine,
                                 init_scale=init_scale, fourier_scale=fourier_scale, image_size=image_size, embedding_type=embedding_type,
                                 input_channels=input_channels, spatial_channels=spatial_channels, dropout=dropout, centered=centered,
                                 discriminative=discriminative, attention=attention, gemm=gemm)
        assert stft is not None, "stft must be provided"          # reference ncsnpp.py:459
        unsupported = []
        if nonlinearity != "swish": unsupported.append("nonlinearity")
        if fir and tuple(int(v) for v in fir_kernel) != (1, 3, 3, 1): unsupported.append("fir_kernel other than (1, 3, 3, 1)")
        if not skip_rescale: unsupported.append("skip_rescale=False")
        if str(resblock_type).lower() != "biggan": unsupported.append("resblock_type")
        if str(progressive).lower() != "output_skip": unsupported.append("progressive")
        if str(progressive_input).lower() != "input_skip": unsupported.append("progressive_input")
        if str(progressive_combine).lower() != "sum": unsupported.append("progressive_combine")
        if str(embedding_type).lower() != "fourier": unsupported.append("embedding_type")
        if input_channels != 2 or spatial_channels != 1: unsupported.append("channels")
        if not time_conditional or discriminative or not centered or dropout not in (0, 0.0):
            unsupported.append("time_conditional/discriminative/centered/dropout")
        res = [image_size // (2 ** i) for i in range(len(ch_mult))]
        if any(r in tuple(attn_resolutions) for r in res): unsupported.append("attn_resolutions")
        if nf % 32: unsupported.append("nf % 32")
        if unsupported:
            raise NotImplementedError("NCSNppTime (MI355X): unsupported options: " + ", ".join(unsupported))
        get = (lambda k: stft[k]) if isinstance(stft, dict) else (lambda k: getattr(stft, k))
        self.stft_kwargs = stft
        self.n_fft, self.hop_length = int(get("n_fft")), int(get("hop_length"))
        assert bool(get("center")), "center=False not supported"
        self.nf, self.ch_mult, self.num_res_blocks = int(nf), tuple(int(c) for c in ch_mult), int(num_res_blocks)
        # attention core: None = library default (BUDDY_ATTN, else "auto": fp32, materialised while T <= 4096, flash beyond); "auto" | "flash" | "bf16" | "f16" | "matrix" (build extension, not a reference key:
        # bf16 / f16 are the opt-in fast mode of DESIGN.md section 4.3)
        if attention is not None and attention not in self.ATTENTION_MODES:
            raise NotImplementedError(f"attention must be one of {sorted(self.ATTENTION_MODES)}")
        self.attention = attention
        # arithmetic of the Winograd-domain GEMMs (build extension): None = library default (BUDDY_GEMM, else "f16x2": power-of-two-scaled two-term
        # f16 split of the fp32 operands, 2^-22, three f16 MFMA products); "bf16x3" = exact three-way bf16 split, six products; "fp32" = v_mfma_f32_32x32x2_f32
        if gemm is not None and gemm not in self.GEMM_MODES:
            raise NotImplementedError(f"gemm must be one of {sorted(self.GEMM_MODES)}")
        self.gemm = gemm
        self._options = {}              # per-handle launcher options set through set_option (build extension; keys: include/buddy_hip.h)
        self.fir = bool(fir)            # FIR (1,3,3,1) resampling instead of nearest / box (reference up_or_down_sampling.py:195-257); no parameters
        self._specs = module_specs(self.nf, self.ch_mult, self.num_res_blocks)
        # The architecture family (reference ncsnpp.py:184-270) is built for any level / block count (fixtures net_cm12_rb2, net_cm1122_rb1), with one
        # restriction of the GroupNorm kernels: they normalise float4 channel quads with one group's statistics, so every normalised tensor -- skip
        # concatenations included -- needs a multiple of 4 channels per group: C <= 128 or C % 128 == 0 (min(C // 4, 32) groups).  nf = 32 / 128 with the
        # reference's multipliers fit; nf = 64, ch_mult (1, 2, 2, 2) has a 192-channel concatenation (six per group) and is refused here.
        bad = sorted({shape[0] for name, shape, kind, _ in self._specs if kind == "gamma" and (shape[0] % 4 or (shape[0] > 128 and shape[0] % 128))})
        if bad:
            raise NotImplementedError(f"NCSNppTime (MI355X): unsupported width nf={self.nf}, ch_mult={self.ch_mult}: GroupNorm over {bad} channels "
                                      "(channels per group must be a multiple of 4: C <= 128 or C % 128 == 0)")
        # parameter containers under the reference names
        n_mod = 1 + max(int(n.split(".")[1]) for n, *_ in self._specs if n.startswith("all_modules."))
        self.all_modules = nn.ModuleList([_Bag() for _ in range(n_mod)])
        self.output_layer = _Bag()
        for name, shape, kind, _ in self._specs:
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if p.isdigit():
                    node = node[int(p)]
                else:
                    if not hasattr(node, p):
                        setattr(node, p, _Bag())
                    node = getattr(node, p)
            node.register_parameter(parts[-1], nn.Parameter(self._init(name, shape, kind, init_scale, fourier_scale),
                                                            requires_grad=False))
        self._handle = None
        self._fwd_id = 0
        self._parent = None

    @staticmethod
    def _init(name, shape, kind, init_scale, fourier_scale):
        if kind == "fourier":
            return torch.randn(*shape) * fourier_scale                      # layerspp.py:37
        if kind in ("b", "beta"):
            return torch.zeros(*shape)
        if kind == "gamma":
            return torch.ones(*shape)
        leaf = name.split(".")[-2] if name.count(".") >= 2 else ""
        scaled = leaf in ("Conv_1", "NIN_3") or (len(shape) == 4 and shape[0] == 2 and shape[2] == 3)
        if name.endswith(".W"):                                             # NIN (in, out): layers.py:551
            return _variance_scaling_uniform(shape, init_scale if scaled else 0.1, 0, 1)
        if name.startswith("output_layer"):
            return _variance_scaling_uniform(shape, 1.0, 1, 0)
        return _variance_scaling_uniform(shape, init_scale if scaled else 1.0, 1, 0)

    # ---- handle management ---------------------------------------------------------------------------
    def _flat_params(self):
        sd = self.state_dict()
        return np.concatenate([sd[n].detach().float().cpu().numpy().ravel() for n, *_ in self._specs]).astype(np.float32)

    def _drop_handle(self):
        if getattr(self, "_handle", None) is not None:
            _lib.load().buddy_ncsnpp_destroy(self._handle)
        self._handle = None

    def _get_handle(self):
        if self._handle is None and getattr(self, "_parent", None) is not None:
            h = C.c_void_p()
            lib = _lib.require_gpu()
            _lib.check(lib.buddy_ncsnpp_replica(self._parent._get_handle(), C.byref(h)))
            if self.attention is not None:      # per-handle setting: a replica may run another attention core on the shared weights
                _lib.check(lib.buddy_ncsnpp_set_attention(h, self.ATTENTION_MODES[self.attention]))
            for k, v in self._options.items():
                _lib.check(lib.buddy_ncsnpp_set_option(h, k.encode(), int(v)))
            self._handle = h
        if self._handle is None:
            lib = _lib.require_gpu()
            blob = np.ascontiguousarray(self._flat_params())
            cm = (C.c_int * len(self.ch_mult))(*self.ch_mult)
            h = C.c_void_p()
            _lib.check(lib.buddy_ncsnpp_create(blob.ctypes.data, blob.size, self.nf, cm, len(self.ch_mult),
                                               self.num_res_blocks, self.n_fft, self.hop_length, C.byref(h)))
            if self.fir:
                _lib.check(lib.buddy_ncsnpp_set_fir(h, 1))
            if self.attention is not None:
                _lib.check(lib.buddy_ncsnpp_set_attention(h, self.ATTENTION_MODES[self.attention]))
            if self.gemm is not None:
                _lib.check(lib.buddy_ncsnpp_set_gemm(h, self.GEMM_MODES[self.gemm]))
            for k, v in self._options.items():
                _lib.check(lib.buddy_ncsnpp_set_option(h, k.encode(), int(v)))
            self._handle = h
        return self._handle

    def set_option(self, key, value):
        """Per-handle launcher option (``buddy_ncsnpp_set_option``: fusion / layout A/B switches, attention core, GEMM arithmetic; an unknown key or
        a value out of range raises).  Applies to this module's handle only -- a replica made afterwards starts from the same settings."""
        lib = _lib.load()
        _lib.check(lib.buddy_option_validate(str(key).encode(), int(value)))     # BEFORE it is remembered: a rejected entry would be replayed (and
        if getattr(self, "_handle", None) is not None:                            # rejected) by every later handle and replica of this module
            _lib.check(lib.buddy_ncsnpp_set_option(self._handle, str(key).encode(), int(value)))
        self._options[str(key)] = int(value)
        return self

    def get_option(self, key):
        v = C.c_int()
        _lib.check(_lib.load().buddy_ncsnpp_get_option(self._get_handle(), str(key).encode(), C.byref(v)))
        return v.value

    def replica(self, attention=None):
        """A second module on the SAME parameters and the same prepared device weights (``buddy_ncsnpp_replica``: reference-counted, read-only)
        with its own library handle -- activation arena + VJP tape -- so another sub-batch can run concurrently on another HIP stream
        (buddy_amd/testing/concurrent.py).  Costs no weight memory and no preparation time.  ``attention``: the replica's attention core (per-handle).  A replica follows the weights its parent had
        when the replica's handle was made; reload the parent -> make new replicas."""
        if attention is not None and attention not in self.ATTENTION_MODES:   # validate BEFORE copying: a half-built copy still holds the parent's handle
            raise NotImplementedError(f"attention must be one of {sorted(self.ATTENTION_MODES)}")
        r = copy.copy(self)                     # shallow: shares _parameters / _modules (the nn.Parameters themselves)
        r.__dict__["_options"] = dict(self._options)
        r.__dict__["_handle"] = None            # first thing: the copy must never own (and on collection destroy) the parent's library handle
        if attention is not None:
            r.attention = attention
        r._fwd_id = 0
        object.__setattr__(r, "_parent", self)   # not a submodule: nn.Module.__setattr__ would register it in the shared _modules
        return r

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._drop_handle()
        self._parent = None                     # new weights: this module owns its handle again
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._drop_handle()
        self._parent = None
        return r

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass

    def weight_bytes(self):
        """Device bytes of the (shared) weight store: {'params', 'packed_1x1', 'conv3_forms', 'conv3_form_count'}."""
        a, b, c, n = C.c_longlong(), C.c_longlong(), C.c_longlong(), C.c_int()
        _lib.check(_lib.require_gpu().buddy_ncsnpp_weight_bytes(self._get_handle(), C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return {"params": a.value, "packed_1x1": b.value, "conv3_forms": c.value, "conv3_form_count": n.value}

    def arena_bytes(self, B, L, with_vjp=True):
        n = C.c_longlong()
        _lib.check(_lib.require_gpu().buddy_ncsnpp_reserve(self._get_handle(), B, L, int(with_vjp), C.byref(n)))
        return n.value

    # ---- compute -------------------------------------------------------------------------------------
    def forward(self, x, time_cond=None):
        """x: (B,1,L) (reference signature) or (B,L); time_cond: (B,) = c_noise."""
        squeeze = x.dim() == 3
        x2 = x[:, 0] if squeeze else x
        y = _NetFn.apply(x2.float(), time_cond.float().reshape(-1), None, self)
        return y[:, None] if squeeze else y

    def denoise_fused(self, x, cnoise, cin, cskip, cout):
        """EDM denoiser with the preconditioning scalars folded into the STFT / overlap-add kernels:
        cskip[b]*x + cout[b]*net(cin[b]*x, cnoise[b]) (reference diff_params/shared.py:98-120).  x: (B,L); rest (B,)."""
        return _NetFn.apply(x.float(), cnoise.float(), (cin.float(), cskip.float(), cout.float()), self)

    def denoise_saved(self, x, scal4):
        """The EDM denoiser (as ``denoise_fused``) WITHOUT an autograd graph: forward with the handle's VJP tape kept; pair it with ``input_vjp``.
        The sampler's fast path (round 6) calls the two library entries directly -- no autograd engine, no ones / sum / mul helper kernels around
        them.  x (B, L); scal4 (4, B) = rows (cnoise, cin, cskip, cout)."""
        lib = _lib.require_gpu()
        x = x.contiguous().float()
        B, L = x.shape
        y = torch.empty_like(x)
        _lib.check(lib.buddy_ncsnpp_forward(self._get_handle(), _lib.ptr(x), _lib.ptr(scal4[0]), _lib.ptr(scal4[1]), _lib.ptr(scal4[2]), _lib.ptr(scal4[3]),
                                            _lib.ptr(y), B, L, 1, _lib.stream_ptr()))
        self._fwd_id += 1           # an autograd node of an earlier forward is stale from here on
        return y

    def input_vjp(self, g):
        """(d denoise_saved / d x)^T g for the last ``denoise_saved`` (``buddy_ncsnpp_vjp``)"""
        g = g.contiguous().float()
        gx = torch.empty_like(g)
        _lib.check(_lib.load().buddy_ncsnpp_vjp(self._get_handle(), _lib.ptr(g), _lib.ptr(gx), _lib.stream_ptr()))
        return gx

    def tap(self, module_idx):
        """(B, frames, bins, C) copy of all_modules[module_idx]'s output from the last forward (parity tests)."""
        p = C.c_void_p()
        dims = (C.c_int * 4)()
        _lib.check(_lib.load().buddy_ncsnpp_tap(self._get_handle(), module_idx, C.byref(p), C.byref(dims)))
        n = int(np.prod(list(dims)))
        out = torch.empty(tuple(dims), dtype=torch.float32, device="cuda")
        _lib.check(_lib.load().buddy_copy_d2d(out.data_ptr(), p, n * 4, _lib.stream_ptr()))
        torch.cuda.synchronize()
        return out
