import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip
from tools.microbench import grid_meta
G = s3d_hip.GridBackend
offs, S, total = grid_meta()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dtype = torch.float16
x = torch.rand(B, 3, device="cuda")
grad = (torch.randn(16, B, 2, device="cuda") * 1e-3).to(dtype)
emb = torch.zeros(total, 2, device="cuda", dtype=dtype)
ge = torch.zeros(total, 2, device="cuda", dtype=dtype)
def run():
    ge.zero_()
    G.grid_encode_backward(grad, x, emb, offs, ge, B, 3, 2, 16, S, 16, None, None, 0, False, 0)
run(); torch.cuda.synchronize(); ref = ge.clone()
print("eager ok", float(ref.float().abs().sum()), flush=True)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    run()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print("side ok", bool(torch.equal(ref, ge)), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    run()
torch.cuda.synchronize(); print("captured", flush=True)
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    print("replay", i, bool(torch.equal(ref, ge)), flush=True)
