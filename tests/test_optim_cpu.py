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


def test_resume_from_a_non_fused_adamw_checkpoint_keeps_this_optimizers_implementation_flags():
    """ADVICE r2 (fit.py:211): a checkpoint of the reference's Lightning run (or of torch.optim.AdamW without fused=True) carries
    fused / capturable / foreach = None / False in its param_groups and `step` counts that torch only casts to float32 on the parameter
    device when the SAVED group says fused or capturable.  Loading it must leave this optimizer what it was constructed as."""
    from ppsurf_amd import optim
    torch.manual_seed(1)
    a = [torch.randn(6, 4, requires_grad=True), torch.randn(9, requires_grad=True)]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    plain = torch.optim.AdamW(b, lr=1e-2, weight_decay=0.1, eps=1e-5)                     # foreach implementation, like the reference's trainer
    for _ in range(3):
        for y in b:
            y.grad = torch.randn_like(y)
        plain.step()
    import copy
    saved = copy.deepcopy(plain.state_dict())          # as read back from a checkpoint file (state_dict() itself shares the tensors)
    assert not saved['param_groups'][0].get('fused') and not saved['param_groups'][0].get('capturable')
    for x, y in zip(a, b):
        x.data.copy_(y.data)
    mine = optim.AdamW(a, lr=1e-2, weight_decay=0.1, eps=1e-5, capturable=False)
    mine.load_state_dict(saved)
    g = mine.param_groups[0]
    assert g['fused'] is True and g['foreach'] is None and g['capturable'] is False
    for p in a:
        st = mine.state[p]
        assert st['step'].dtype == torch.float32 and st['step'].device == p.device and float(st['step']) == 3.0
    for x, y in zip(a, b):
        gr = torch.randn_like(x)
        x.grad, y.grad = gr.clone(), gr.clone()
    mine.step()
    plain.step()
    for x, y in zip(a, b):
        torch.testing.assert_close(x, y, rtol=1e-6, atol=1e-7)
