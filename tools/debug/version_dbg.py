import torch
p = torch.nn.Parameter(torch.randn(1000, device="cuda"))
q = torch.nn.Parameter(torch.randn(1000, device="cuda"))
opt = torch.optim.Adam([{"params": [p]}, {"params": [q]}], lr=1e-2, fused=True)
sc = torch.amp.GradScaler("cuda")
for i in range(3):
    opt.zero_grad()
    loss = (p ** 2).sum() + (q ** 2).sum()
    sc.scale(loss).backward()
    v0 = (p._version, q._version)
    sc.step(opt); sc.update()
    print("step", i, "version before", v0, "after", (p._version, q._version), float(p.detach().abs().sum()))
