"""Synthetic weights and inputs for parity tests and benchmarks.

No pretrained checkpoint is available offline (reference ``README.md:13``) and the reference's default
initialisation makes half of the network numerically invisible (``init_scale: 0`` -> 1e-10 variance,
reference ``networks/ncsnpp_utils/layers.py:88-91``), so parity is established on identical
*re-randomised* weights: a frozen ``numpy.random.RandomState`` bit-stream, fan-in scaled, regenerated on
both sides.  Parameter names and shapes are the reference ``state_dict`` ones
(``all_modules.N.*``, ``output_layer.*``).
"""
from __future__ import annotations

import numpy as np


def module_specs(nf=128, ch_mult=(1, 2, 2, 2), num_res_blocks=1, in_ch=2):
    """Ordered (name, shape, kind, fan_in) list for the shipped architecture family
    (biggan resblocks, input_skip/sum, output_skip, one bottleneck attention;
    construction order = reference ``networks/ncsnpp.py:157-274``)."""
    specs = []
    idx = [0]

    def add(suffix, shape, kind, fan_in=1):
        specs.append((f"all_modules.{idx[0]}.{suffix}", tuple(shape), kind, fan_in))

    def nxt():
        idx[0] += 1

    def resblock(cin, cout, resample=False):
        add("GroupNorm_0.weight", (cin,), "gamma"); add("GroupNorm_0.bias", (cin,), "beta")
        add("Conv_0.weight", (cout, cin, 3, 3), "w", cin * 9); add("Conv_0.bias", (cout,), "b")
        add("Dense_0.weight", (cout, nf * 4), "w", nf * 4); add("Dense_0.bias", (cout,), "b")
        add("GroupNorm_1.weight", (cout,), "gamma"); add("GroupNorm_1.bias", (cout,), "beta")
        add("Conv_1.weight", (cout, cout, 3, 3), "w", cout * 9); add("Conv_1.bias", (cout,), "b")
        if cin != cout or resample:
            add("Conv_2.weight", (cout, cin, 1, 1), "w", cin); add("Conv_2.bias", (cout,), "b")
        nxt()

    add("W", (nf,), "fourier"); nxt()
    add("weight", (nf * 4, nf * 2), "w", nf * 2); add("bias", (nf * 4,), "b"); nxt()
    add("weight", (nf * 4, nf * 4), "w", nf * 4); add("bias", (nf * 4,), "b"); nxt()
    add("weight", (nf, in_ch, 3, 3), "w", in_ch * 9); add("bias", (nf,), "b"); nxt()
    hs_c = [nf]
    c = nf
    nres = len(ch_mult)
    for lvl in range(nres):
        for _ in range(num_res_blocks):
            co = nf * ch_mult[lvl]
            resblock(c, co); c = co
            hs_c.append(c)
        if lvl != nres - 1:
            resblock(c, c, resample=True)
            add("Conv_0.weight", (c, in_ch, 1, 1), "w", in_ch); add("Conv_0.bias", (c,), "b"); nxt()
            hs_c.append(c)
    resblock(c, c)
    add("GroupNorm_0.weight", (c,), "gamma"); add("GroupNorm_0.bias", (c,), "beta")
    for k in range(4):
        add(f"NIN_{k}.W", (c, c), "w", c); add(f"NIN_{k}.b", (c,), "b")
    nxt()
    resblock(c, c)
    for lvl in reversed(range(nres)):
        for _ in range(num_res_blocks + 1):
            co = nf * ch_mult[lvl]
            resblock(c + hs_c.pop(), co); c = co
        add("weight", (c,), "gamma"); add("bias", (c,), "beta"); nxt()
        add("weight", (in_ch, c, 3, 3), "w", c * 9); add("bias", (in_ch,), "b"); nxt()
        if lvl != 0:
            resblock(c, c, resample=True)
    assert not hs_c
    specs.append(("output_layer.weight", (2, in_ch, 1, 1), "w", in_ch))
    specs.append(("output_layer.bias", (2,), "b", 1))
    return specs


def synth_state_dict(seed=0, nf=128, ch_mult=(1, 2, 2, 2), num_res_blocks=1, fourier_scale=16.0):
    """Deterministic float32 weights keyed by reference state-dict names (numpy arrays)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, shape, kind, fan_in in module_specs(nf, ch_mult, num_res_blocks):
        if kind == "w":
            a = rs.standard_normal(shape) / np.sqrt(fan_in)
        elif kind == "b":
            a = 0.1 * rs.standard_normal(shape)
        elif kind == "gamma":
            a = 1.0 + 0.1 * rs.standard_normal(shape)
        elif kind == "beta":
            a = 0.1 * rs.standard_normal(shape)
        elif kind == "fourier":
            a = fourier_scale * rs.standard_normal(shape)
        else:
            raise ValueError(kind)
        sd[name] = a.astype(np.float32)
    return sd


def synth_clean(utt_id, length=64000, sigma_data=0.05):
    """Synthetic "clean" utterance, normalised like the harness does (reference ``testing/tester.py:134-135``).
    Speech-like enough for plumbing: coloured noise with a slow envelope."""
    rs = np.random.RandomState(1234 + utt_id)
    n = rs.standard_normal(length + 64)
    k = np.exp(-np.arange(64) / 8.0)
    x = np.convolve(n, k, mode="full")[64:64 + length]
    env = 0.55 + 0.45 * np.sin(2 * np.pi * np.arange(length) / 16000.0 * (1.5 + 0.1 * utt_id))
    x = x * env
    x = sigma_data * x / x.std(ddof=1)
    return x.astype(np.float32)


def synth_rir(utt_id, taps=8000, t60=0.5, fs=16000):
    """exp-decay noise RIR, direct path first and peak-normalised (mirrors reference ``datasets/vctk.py:211-214``)."""
    rs = np.random.RandomState(4321 + utt_id)
    t = np.arange(taps) / fs
    h = np.exp(-6.908 * t / t60) * rs.standard_normal(taps) * 0.3
    h[0] = 1.0
    h = h[np.argmax(np.abs(h)):]
    h = h / np.abs(h).max()
    return h.astype(np.float32)


def synth_noise(utt_id, n_draws, length):
    """Pre-drawn N(0,1) noise, one row per sampler draw (injected so fixtures are exact; the reference draws
    from the torch CPU generator, ``testing/EulerHeunSampler.py:21,43``)."""
    rs = np.random.RandomState(777 + utt_id)
    return rs.standard_normal((n_draws, length)).astype(np.float32)
