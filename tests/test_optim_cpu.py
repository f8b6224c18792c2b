"""ppsurf_amd.optim.AdamW off the GPU: it is torch's fused AdamW (the HIP step only takes device tensors)."""
import torch


def test_cpu_parameters_take_torchs_own_step():
    from ppsurf_amd import optim
    torch.manual_seed(0)
    a = [torch.randn(7, 5, requires_grad=True), torch.randn(33, requires_grad=True)]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    mine, ref = optim.AdamW(a, lr=1e-2, weight_decay=0.1, eps=1e-5), torch.optim.AdamW(b, lr=1e-2, weight_decay=0.1, eps=1e-5)
    for step in range(4):
        for x, y in zip(a, b):
            g = torch.randn_like(x)
            x.grad, y.grad = g.clone(), g.clone()
        mine.step()
        ref.step()
    assert mine.fast_steps == 0
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=1e-6, atol=1e-7)
    sd = mine.state_dict()
    assert set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'} and float(sd['state'][0]['step']) == 4.0
