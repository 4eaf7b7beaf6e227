import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
from gridencoder import GridEncoder
from nerf.optim import NativeAdam, NativeGradScaler
torch.manual_seed(0)
ea = GridEncoder(num_levels=4, base_resolution=4, log2_hashmap_size=10, desired_resolution=32).cuda()
eb = GridEncoder(num_levels=4, base_resolution=4, log2_hashmap_size=10, desired_resolution=32).cuda()
eb.load_state_dict(ea.state_dict())
w = torch.randn(8, 1, generator=torch.Generator().manual_seed(3)).cuda()
oa = NativeAdam([{"params": ea.parameters()}], lr=1e-2)
sa = NativeGradScaler("cuda", init_scale=1024.0)
ob = torch.optim.Adam(eb.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
sb = torch.amp.GradScaler("cuda", init_scale=1024.0)
for step in range(4):
    x = (torch.rand(9000, 3, generator=torch.Generator().manual_seed(step)) * 2 - 1).cuda()
    gs = []
    for enc, opt, sc in ((ea, oa, sa), (eb, ob, sb)):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            y = enc(x, bound=1)
            loss = ((y.float() @ w) ** 2).mean()
        sc.scale(loss).backward()
        gs.append(enc.embeddings._s3d_grad.float().clone() if enc is ea else enc.embeddings.grad.clone())
        sc.step(opt)
        sc.update()
    d = (ea.embeddings.detach() - eb.embeddings.detach()).abs()
    i = int(d.flatten().argmax())
    r, c = divmod(i, 2)
    print(f"step {step}: grad equal {bool(torch.equal(gs[0], gs[1]))} max |dp| {float(d.max()):.3e} at {r},{c}: pa {float(ea.embeddings[r,c]):.6e} pb {float(eb.embeddings[r,c]):.6e} "
          f"g {float(gs[0][r,c]):.4e} m {float(oa.state[ea.embeddings]['exp_avg'][r,c]):.4e}/{float(ob.state[eb.embeddings]['exp_avg'][r,c]):.4e} "
          f"v {float(oa.state[ea.embeddings]['exp_avg_sq'][r,c]):.4e}/{float(ob.state[eb.embeddings]['exp_avg_sq'][r,c]):.4e}")
