cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_optim_gpu.py -m gpu -q -p no:cacheprovider --tb=line 2>&1 | tail -8
python - <<'PY'
import torch, dreamgaussian_amd as D
dev = torch.device("cuda:0")
def mk(cls):
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5000, 3, device=dev)), torch.nn.Parameter(torch.randn(5000, 1, device=dev))]
    return ps, cls([{"params": [ps[0]], "lr": 1e-3}, {"params": [ps[1]], "lr": 5e-2}], lr=0.0, eps=1e-15)
pa, oa = mk(D.FusedAdam); pb, ob = mk(torch.optim.Adam)
g = torch.Generator().manual_seed(1)
for s in range(5):
    for x, y in zip(pa, pb):
        gr = torch.randn(x.shape, generator=g).to(dev); x.grad = gr.clone(); y.grad = gr.clone()
    oa.step(); ob.step()
    for x, y in zip(pa, pb):
        sa, sb = oa.state[x], ob.state[y]
        print(s, "param max abs diff", (x - y).abs().max().item(), "exp_avg", (sa["exp_avg"] - sb["exp_avg"]).abs().max().item(), "exp_avg_sq", (sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max().item(), "bitwise equal:", torch.equal(x, y))
PY
