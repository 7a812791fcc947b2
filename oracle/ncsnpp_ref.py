"""Oracle: NCSN++ score network over STFT spectrograms, functional PyTorch fp32 on CPU.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates reference ``networks/ncsnpp.py:281-449``
(U-Net graph), ``:473-506`` (STFT / iSTFT wrapper) and the live layers of
``networks/ncsnpp_utils/layerspp.py`` for the shipped architecture family
(``conf/network/ncsnpp.yaml``: biggan resblocks, input_skip/sum, output_skip, fir=False, one
bottleneck attention, skip_rescale).  Parameters are a dict keyed by reference state-dict names.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def _t(P, name):
    v = P[name]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def group_norm(P, prefix, x):
    """nn.GroupNorm(min(C//4, 32), C, eps=1e-6) -- reference layerspp.py:219,231; ncsnpp.py:239,251."""
    C = x.shape[1]
    return F.group_norm(x, min(C // 4, 32), _t(P, prefix + ".weight"), _t(P, prefix + ".bias"), eps=1e-6)


def conv(P, prefix, x, pad):
    """ddpm_conv3x3 / ddpm_conv1x1 -- reference layers.py:100-106,119-126."""
    return F.conv2d(x, _t(P, prefix + ".weight"), _t(P, prefix + ".bias"), padding=pad)


def nin(P, prefix, x):
    """NIN: per-pixel x @ W + b with W stored (in, out) -- reference layers.py:548-557."""
    y = torch.einsum("bchw,cd->bdhw", x, _t(P, prefix + ".W"))
    return y + _t(P, prefix + ".b")[None, :, None, None]


def up2(x):
    """naive_upsample_2d: nearest x2 -- reference up_or_down_sampling.py:59-63."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def down2(x):
    """naive_downsample_2d: 2x2 mean -- reference up_or_down_sampling.py:66-69."""
    B, C, H, W = x.shape
    return x.reshape(B, C, H // 2, 2, W // 2, 2).mean(dim=(3, 5))


def _upfirdn(x, k2, up, down, pad0, pad1):
    """upfirdn2d_native -- reference op/upfirdn2d.py:171-215 (zero-insert by ``up``, zero-pad, correlate with the flipped kernel, decimate),
    the function the reference's dispatcher runs on CPU (:145-156)."""
    B, C, H, W = x.shape
    z = x.reshape(B * C, 1, H, 1, W, 1)
    z = F.pad(z, [0, up - 1, 0, 0, 0, up - 1]).reshape(B * C, 1, H * up, W * up)
    z = F.pad(z, [pad0, pad1, pad0, pad1])
    z = F.conv2d(z, torch.flip(k2, [0, 1])[None, None])
    return z[:, :, ::down, ::down].reshape(B, C, z.shape[2] // down + (z.shape[2] % down > 0), -1)


def _fir_kernel(k=(1, 3, 3, 1)):
    k1 = torch.tensor(k, dtype=torch.get_default_dtype())
    k2 = torch.outer(k1, k1)
    return k2 / k2.sum()


def fir_up2(x, k=(1, 3, 3, 1)):
    """upsample_2d(x, k, factor=2) -- reference up_or_down_sampling.py:195-224: kernel * factor^2, pad ((p+1)//2 + factor - 1, p//2), p = len(k) - factor."""
    p = len(k) - 2
    return _upfirdn(x, _fir_kernel(k) * 4, 2, 1, (p + 1) // 2 + 1, p // 2)


def fir_down2(x, k=(1, 3, 3, 1)):
    """downsample_2d(x, k, factor=2) -- reference up_or_down_sampling.py:227-257: pad ((p+1)//2, p//2)."""
    p = len(k) - 2
    return _upfirdn(x, _fir_kernel(k), 1, 2, (p + 1) // 2, p // 2)


def resblock(P, m, x, temb, mode=None, fir=False):
    """ResnetBlockBigGANpp.forward -- reference layerspp.py:242-274 (fir=True: FIR resampling of both h and x, :246-255)."""
    pre = f"all_modules.{m}"
    h = F.silu(group_norm(P, pre + ".GroupNorm_0", x))
    if mode == "up":
        h, x = (fir_up2(h), fir_up2(x)) if fir else (up2(h), up2(x))
    elif mode == "down":
        h, x = (fir_down2(h), fir_down2(x)) if fir else (down2(h), down2(x))
    h = conv(P, pre + ".Conv_0", h, 1)
    h = h + F.linear(F.silu(temb), _t(P, pre + ".Dense_0.weight"), _t(P, pre + ".Dense_0.bias"))[:, :, None, None]
    h = F.silu(group_norm(P, pre + ".GroupNorm_1", h))
    h = conv(P, pre + ".Conv_1", h, 1)
    if (pre + ".Conv_2.weight") in P:
        x = conv(P, pre + ".Conv_2", x, 0)
    return (x + h) / SQRT2


def attnblock(P, m, x):
    """AttnBlockpp.forward -- reference layerspp.py:75-91 (single head over all H*W positions)."""
    pre = f"all_modules.{m}"
    B, C, H, W = x.shape
    h = group_norm(P, pre + ".GroupNorm_0", x)
    q = nin(P, pre + ".NIN_0", h).reshape(B, C, H * W)
    k = nin(P, pre + ".NIN_1", h).reshape(B, C, H * W)
    v = nin(P, pre + ".NIN_2", h).reshape(B, C, H * W)
    w = torch.einsum("bci,bcj->bij", q, k) * (int(C) ** (-0.5))
    w = F.softmax(w, dim=-1)
    h = torch.einsum("bij,bcj->bci", w, v).reshape(B, C, H, W)
    h = nin(P, pre + ".NIN_3", h)
    return (x + h) / SQRT2


def time_embedding(P, cnoise):
    """GaussianFourierProjection + 2 Linear -- reference layerspp.py:39-41, ncsnpp.py:299-318."""
    proj = cnoise[:, None] * _t(P, "all_modules.0.W")[None, :] * 2 * math.pi
    temb = torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)
    temb = F.linear(temb, _t(P, "all_modules.1.weight"), _t(P, "all_modules.1.bias"))
    temb = F.linear(F.silu(temb), _t(P, "all_modules.2.weight"), _t(P, "all_modules.2.bias"))
    return temb


def unet(P, x, cnoise, ch_mult=(1, 2, 2, 2), num_res_blocks=1, taps=None, fir=False):
    """NCSNpp.forward on real/imag channels -- reference ncsnpp.py:281-449.
    x: (B, 2, F, T) real.  Returns (B, 2, F, T).  ``taps`` (optional dict) records intermediate
    tensors by all_modules index for per-module parity checks."""
    nres = len(ch_mult)
    temb = time_embedding(P, cnoise)
    m = 3
    pyr_in = x
    hs = [conv(P, f"all_modules.{m}", x, 1)]
    m += 1

    def rec(i, t):
        if taps is not None:
            taps[i] = t

    rec(3, hs[0])
    for lvl in range(nres):
        for _ in range(num_res_blocks):
            h = resblock(P, m, hs[-1], temb); rec(m, h); m += 1
            hs.append(h)
        if lvl != nres - 1:
            h = resblock(P, m, hs[-1], temb, "down", fir); rec(m, h); m += 1
            pyr_in = fir_down2(pyr_in) if fir else F.avg_pool2d(pyr_in, 2, stride=2)   # layerspp.py:159 / :156
            h = conv(P, f"all_modules.{m}.Conv_0", pyr_in, 0) + h           # Combine(sum), layerspp.py:52-57
            rec(m, h); m += 1
            hs.append(h)
    h = hs[-1]
    h = resblock(P, m, h, temb); rec(m, h); m += 1
    h = attnblock(P, m, h); rec(m, h); m += 1
    h = resblock(P, m, h, temb); rec(m, h); m += 1
    pyr = None
    for lvl in reversed(range(nres)):
        for _ in range(num_res_blocks + 1):
            h = resblock(P, m, torch.cat([h, hs.pop()], dim=1), temb); rec(m, h); m += 1
        ph = F.silu(group_norm(P, f"all_modules.{m}", h)); m += 1
        ph = conv(P, f"all_modules.{m}", ph, 1); m += 1
        if pyr is None:
            pyr = ph
        else:
            pyr = (fir_up2(pyr) if fir else F.interpolate(pyr, scale_factor=2, mode="nearest")) + ph   # layerspp.py:122 / :117
        rec(m - 1, pyr)
        if lvl != 0:
            h = resblock(P, m, h, temb, "up", fir); rec(m, h); m += 1
    assert not hs
    out = F.conv2d(pyr, _t(P, "output_layer.weight"), _t(P, "output_layer.bias"))
    return out


def stft(sig, n_fft=510, hop=128):
    """NCSNppTime.stft -- reference ncsnpp.py:473-486. sig (B,L) -> complex (B, n_fft//2+1, T') with frames
    zero-padded to a multiple of 16."""
    win = torch.hann_window(n_fft, periodic=True)
    spec = torch.stft(sig, n_fft=n_fft, hop_length=hop, window=win, center=True, return_complex=True)
    T = spec.shape[-1]
    if T % 16 != 0:
        spec = F.pad(spec, (0, 16 - T % 16))
    return spec


def istft(spec, length, n_fft=510, hop=128):
    """NCSNppTime.istft -- reference ncsnpp.py:489-496."""
    win = torch.hann_window(n_fft, periodic=True)
    return torch.istft(spec, n_fft=n_fft, hop_length=hop, window=win, center=True, length=length)[..., :length]


def ncsnpp_time(P, x, cnoise, n_fft=510, hop=128, ch_mult=(1, 2, 2, 2), num_res_blocks=1, taps=None, fir=False):
    """NCSNppTime.forward -- reference ncsnpp.py:498-506.  x: (B,1,L) or (B,L); returns same shape."""
    squeeze = x.dim() == 3
    sig = x[:, 0] if squeeze else x
    L = sig.shape[-1]
    S = stft(sig, n_fft, hop)
    xri = torch.stack([S.real, S.imag], dim=1)                # ncsnpp.py:291-297
    o = unet(P, xri, cnoise, ch_mult, num_res_blocks, taps, fir)
    So = torch.complex(o[:, 0].contiguous(), o[:, 1].contiguous())  # ncsnpp.py:446-448
    y = istft(So, L, n_fft, hop)
    return y[:, None] if squeeze else y


def to_torch(sd):
    """state dict (numpy or torch) -> torch tensors of the default dtype (fp32; fp64 inside ``oracle.precision.fp64()``)"""
    dt, dev = torch.get_default_dtype(), torch.get_default_device()
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)).to(device=dev, dtype=dt) for k, v in sd.items()}
