"""Long-form policy (BASELINE config 5; not in the reference, which samples one ``(1, L)`` clip per call, testing/tester.py:153).

Default = UN-CHUNKED: the whole utterance goes through the sampler in one piece -- with the flash attention kernel the 30 s case needs
no 905 MB attention matrix and fits the 288 GB of one MI355X many times over (profiles/DESIGN_history_r01-r04.md section 5, "Long form").  For inputs beyond what one GPU
should hold, or to turn one very long recording into a batch (utterance-parallel over chunks, also across ranks), the clip is cut into
equal-length, overlapping chunks that are sampled as independent utterances (own noise stream, own blind operator / RIR estimate) and
cross-faded back with linear ramps over the overlaps (a partition of unity: chunking the identity returns the input exactly)."""
from __future__ import annotations

import math

import torch


def chunk_plan(L, chunk, overlap):
    """equal-length chunks covering [0, L): starts (first 0, last L - chunk) spaced so that neighbours overlap by >= ``overlap`` samples"""
    if L <= chunk:
        return [0], L
    assert 0 <= overlap < chunk, "overlap must be smaller than the chunk"
    n = max(2, math.ceil((L - overlap) / (chunk - overlap)))
    return [round(i * (L - chunk) / (n - 1)) for i in range(n)], chunk


def crossfade_weights(starts, chunk, L, device=None):
    """(n, chunk) weights: 1 in the interior, linear ramps over the overlap with each neighbour; columns sum to 1 at every sample"""
    n = len(starts)
    w = torch.ones(n, chunk, device=device)
    for i in range(n - 1):
        ov = starts[i] + chunk - starts[i + 1]
        assert ov >= 0, "chunks must not leave gaps"
        if ov == 0:                       # L an exact multiple of the chunk with overlap 0: neighbours abut, plain concatenation
            continue
        ramp = (torch.arange(ov, device=device, dtype=torch.float32) + 0.5) / ov
        w[i, chunk - ov:] *= 1.0 - ramp
        w[i + 1, :ov] *= ramp
    return w


def split(y, chunk, overlap):
    """y (L,) or (1, L) -> (chunks (n, chunk), starts)"""
    y = y.reshape(-1)
    starts, clen = chunk_plan(y.shape[-1], chunk, overlap)
    return torch.stack([y[s:s + clen] for s in starts]), starts


def merge(parts, starts, L):
    """overlap-add of (n, chunk) sampled chunks back to (L,)"""
    n, chunk = parts.shape
    if n == 1:
        return parts[0, :L]
    w = crossfade_weights(starts, chunk, L, device=parts.device)
    out = torch.zeros(L, device=parts.device, dtype=parts.dtype)
    for i, s in enumerate(starts):
        out[s:s + chunk] += w[i] * parts[i]
    return out


def predict_chunked(sample_batch, y, chunk, overlap, level_match=False):
    """``sample_batch``: (n, chunk) reverberant chunks -> (n, chunk) estimates (one sampler call, chunks = utterances).

    ``level_match``: the blind configuration rescales every utterance's estimate to a fixed standard deviation
    (``constraint_speech_magnitude``, reference EulerHeunSamplerDPS.py:127-129) -- per CHUNK here, so a chunk that is mostly a pause would come
    back as loud as a chunk of running speech and the cross-fade would mix segments of different gains.  With ``level_match`` each chunk's
    estimate is scaled by std(y_chunk) / std(y_clip) (the observation's own level profile) before the merge, which restores one gain for the clip;
    what remains chunk-specific is the RIR estimate (one operator per chunk), see profiles/DESIGN_history_r01-r04.md section 5.  Residual bias: the gain profile is the
    OBSERVATION's, and reverberation fills pauses, so a pause comes back louder than in the clean signal (measured pause / speech level 0.107
    against 0.052 in the input of tests/test_hip_cli.py's clip).  Chunks are never padded (equal chunks, the last one starts at L - chunk), so
    every std is over valid samples."""
    parts, starts = split(y, chunk, overlap)
    est = sample_batch(parts)
    if level_match and parts.shape[0] > 1:
        g = parts.std(dim=1, keepdim=True) / (y.reshape(-1).std() + 1e-12)
        est = est * g.to(est.dtype)
    return merge(est, starts, y.reshape(-1).shape[-1])
