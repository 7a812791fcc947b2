"""Long-form parity (BASELINE.json configs[4]: 30 s @ 16 kHz = 480 000 samples, the maximum size the reference documents):
network forward + input-VJP of one utterance on the MI355X path vs the CPU oracle, plus one blind DPS step.
usage: python tools/validate_longform.py [L]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_hip_network import build
from buddy_amd.synth import synth_state_dict
from oracle import ncsnpp_ref

L = int(sys.argv[1]) if len(sys.argv) > 1 else 480000
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())
net = build(128, 510, 128, 0)
rs = np.random.RandomState(11)
x = torch.from_numpy((0.4 * rs.standard_normal((1, L))).astype(np.float32))
cot = torch.from_numpy(rs.standard_normal((1, L)).astype(np.float32))
cn = torch.tensor([-0.6])
xg = x.cuda().requires_grad_(True)
y = net(xg, cn.cuda()); g, = torch.autograd.grad(y, xg, cot.cuda())
torch.cuda.synchronize(); t0 = time.time()
y = net(xg, cn.cuda()); g, = torch.autograd.grad(y, xg, cot.cuda())
torch.cuda.synchronize(); tg = time.time() - t0
torch.set_num_threads(32)
P = ncsnpp_ref.to_torch(synth_state_dict(0, 128))
t0 = time.time()
xr = x.clone().requires_grad_(True)
yr = ncsnpp_ref.ncsnpp_time(P, xr, cn, 510, 128)
gr, = torch.autograd.grad(yr, xr, cot)
tc = time.time() - t0
print(json.dumps({"L": L, "rel_err_forward": rel(y.detach(), yr.detach()), "rel_err_vjp": rel(g, gr), "gpu_ms_fwd_vjp": 1e3 * tg,
                  "oracle_s_fwd_vjp_32_threads": tc, "arena_GB": net.arena_bytes(1, L, True) / 1e9}))
