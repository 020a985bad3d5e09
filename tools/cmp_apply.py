import sys, torch
a = torch.load(sys.argv[1]); b = torch.load(sys.argv[2])
worst = 0
for (i, ya, dxa, ra, ka), (j, yb, dxb, rb, kb) in zip(a, b):
    for name, u, v in (("y", ya, yb), ("dx", dxa, dxb), ("rows", ra, rb), ("kb", ka, kb)):
        if u is None: continue
        d = (u - v).abs().max().item(); m = v.abs().max().item()
        if d > 0: print(i, name, "max diff %.3e of max %.3e (rel %.2e)" % (d, m, d / max(m, 1e-30)))
        worst = max(worst, d / max(m, 1e-30))
print("worst rel", worst)
