"""AdamW of the fit loop with the whole step as ONE HIP launch (pps_optim.hip).

`ppsurf_amd.optim.AdamW` IS a `torch.optim.AdamW(fused=True)`: same constructor, same state (`step`, `exp_avg`, `exp_avg_sq` per parameter, so
`state_dict()` / `load_state_dict()` and the checkpoints written by ppsurf_amd.fit keep the reference's layout), same interplay with
`torch.amp.GradScaler` (`grad_scale` / `found_inf` are honoured inside the kernel, nothing is read back) and with HIP-graph capture
(`capturable=True`: learning rate and step counts live on the device).  Only `step()` differs: instead of torch's multi-tensor kernels
(nine launches, 0.78 ms for the 298 tensors / 13.7 M parameters of PPSurf) it launches one kernel over a device table of 4096-element
pieces of all parameters (0.064 ms).  The table holds raw pointers, so it is rebuilt whenever a parameter, gradient or state tensor moved;
ppsurf_amd.fit keeps the gradients in the flat buffers of sharding.GradBuckets, where they do not move (a caller whose gradients are allocated
anew by every backward pass is handed to torch's step after eight rebuilds in a row: the host-side rebuild costs more than it saves).  Whatever the fast path does not
take (CPU tensors, non-fp32 or non-contiguous parameters, amsgrad / maximize, a table that would have to be rebuilt while a graph is being
captured) goes through torch's own step.

replaces: the `optimizer:` of configs/poco.yaml:60-69 as the reference's trainer steps it.
"""
import numpy as np
import torch

from . import _lib

PIECE = 4096                 # elements per workgroup


class AdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False, foreach=None,
                 capturable=False, differentiable=False, fused=None):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize, foreach=None,
                         capturable=capturable, differentiable=differentiable, fused=True)
        self._tables = {}        # group index -> (signature, pieces tensor, step-pointer tensor, n_pieces, n_steps, quick key)
        self._moved = {}         # group index -> consecutive steps that had to rebuild its table (gradients allocated anew by every backward pass)
        self.fast_steps = 0      # steps taken by the HIP kernel (tests / diagnostics)

    # ---- fast path ------------------------------------------------------------------------------------------------------------------
    def _fast_params(self, group):
        """Parameters of the group that have a gradient, if the HIP kernel can take ALL of them; else None."""
        if group['amsgrad'] or group['maximize'] or group.get('differentiable'):
            return None
        lr = group['lr']
        if torch.is_tensor(lr) and lr.is_cuda and (lr.dtype != torch.float32 or lr.numel() != 1):
            return None
        ps = [p for p in group['params'] if p.grad is not None]
        for p in ps:
            g = p.grad
            if (not p.is_cuda or p.dtype != torch.float32 or g.dtype != torch.float32 or g.is_sparse or not p.is_contiguous()
                    or not g.is_contiguous() or g.device != p.device):
                return None
        if len({p.device for p in ps}) > 1:                          # one table, one launch: a group spread over devices is torch's business
            return None
        return ps

    def _init_state(self, ps):
        fresh = [p for p in ps if len(self.state[p]) == 0]
        if not fresh:
            return
        self._tables.clear()                                          # new state tensors: no cached table may point at what these replace
        steps = torch.zeros((len(fresh),), dtype=torch.float32, device=fresh[0].device)     # one allocation for the step counts of the batch
        for i, p in enumerate(fresh):
            st = self.state[p]
            st['step'] = steps[i]
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _table(self, gi, ps):
        # the table holds raw pointers: it is valid only while parameters, gradients AND state tensors are where they were (a caller may
        # clear or reassign optimizer.state without going through load_state_dict)
        state = self.state
        quick = tuple((p.data_ptr(), p.grad.data_ptr(), state[p]['exp_avg'].data_ptr(), state[p]['exp_avg_sq'].data_ptr(), state[p]['step'].data_ptr())
                      for p in ps)
        cached = self._tables.get(gi)
        if cached is not None and cached[5] == quick:
            self._moved[gi] = 0
            return cached
        if cached is not None:
            self._moved[gi] = self._moved.get(gi, 0) + 1
            if self._moved[gi] > 8:
                return None                                          # gradients never stay put (no flat gradient buffers): building a 3600-row
                                                                     # table on the host every step costs more than torch's nine launches
        sig, rows, step_ptrs = [], [], []
        for p in ps:
            st = self.state[p]
            m, v, s = st['exp_avg'], st['exp_avg_sq'], st['step']
            if (not m.is_contiguous() or not v.is_contiguous() or m.dtype != torch.float32 or v.dtype != torch.float32 or not s.is_cuda
                    or s.dtype != torch.float32 or s.numel() != 1):
                return None
            sig.append((p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), s.data_ptr(), p.numel()))
        sig = tuple(sig)
        if torch.cuda.is_current_stream_capturing():
            return None                                              # a host-to-device copy cannot be recorded: torch's step for this capture
        for pp, gp, mp, vp, sp, n in sig:
            step_ptrs.append(sp)
            for off in range(0, n, PIECE):
                rows.append((pp + 4 * off, gp + 4 * off, mp + 4 * off, vp + 4 * off, sp, min(PIECE, n - off)))
        piece_bytes = _lib.lib().pps_adamw_piece_bytes()
        assert piece_bytes == 48
        dev = ps[0].device
        pieces = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)       # [n, 6] int64: four pointers, the step pointer, (n | pad << 32)
        steps = torch.from_numpy(np.asarray(step_ptrs, dtype=np.int64)).to(dev)
        cached = (sig, pieces, steps, len(rows), len(step_ptrs), quick)
        self._tables[gi] = cached
        return cached

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        plan = []
        for gi, group in enumerate(self.param_groups):
            ps = self._fast_params(group)
            if ps is None:
                plan = None
                break
            if not ps:
                continue
            self._init_state(ps)
            table = self._table(gi, ps)
            if table is None:
                plan = None
                break
            plan.append((group, table))
        if plan is None:
            super().step()                                            # torch's fused multi-tensor step (same arithmetic, same state)
            return loss
        grad_scale, found_inf = getattr(self, 'grad_scale', None), getattr(self, 'found_inf', None)
        lib = _lib.lib()
        for group, (_, pieces, steps, n_pieces, n_steps, _q) in plan:
            lr = group['lr']
            lr_dev = lr if torch.is_tensor(lr) and lr.is_cuda else None
            beta1, beta2 = group['betas']
            dev = pieces.device
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                _lib.check(lib.pps_adamw_step(pieces.data_ptr(), n_pieces, steps.data_ptr(), n_steps,
                                              lr_dev.data_ptr() if lr_dev is not None else None, 0.0 if lr_dev is not None else float(lr),
                                              float(beta1), float(beta2), float(group['eps']), float(group['weight_decay']),
                                              grad_scale.data_ptr() if grad_scale is not None else None,
                                              found_inf.data_ptr() if found_inf is not None else None, stream), 'pps_adamw_step')
        self.fast_steps += 1
        return loss

    def load_state_dict(self, state_dict):
        """Accepts the state of ANY AdamW of the same parameters -- in particular a checkpoint written by the reference's Lightning run or by a
        non-fused torch.optim.AdamW (`fused` / `capturable` / `foreach` None or False in its param_groups, `step` counts saved as CPU scalars).
        torch's load_state_dict takes those three keys from the SAVED groups, both for the groups themselves and for the decision whether the
        `step` tensors follow the parameters to the device as float32; with the saved values this optimizer would come back as a foreach
        implementation with host-side step counts, which neither GradScaler's fused hand-over nor graph capture accepts.  The implementation
        flags are a property of this object, not of the checkpoint: they are put in front of the saved ones."""
        own = self.param_groups
        saved = state_dict['param_groups']
        if len(saved) != len(own):
            raise ValueError('loaded state dict has a different number of parameter groups')
        patched = [dict(g, fused=o.get('fused'), capturable=o.get('capturable', False), foreach=o.get('foreach'), differentiable=o.get('differentiable', False))
                   for g, o in zip(saved, own)]
        super().load_state_dict({'state': state_dict['state'], 'param_groups': patched})
        self._tables.clear()                                          # the state tensors were replaced
        self._moved.clear()
