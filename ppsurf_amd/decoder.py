"""Host side of the fused occupancy decoder (PPSurfNetwork.from_latent, source/ppsurf_model.py:82-117).

`DecoderPlan` turns the reference's state dict (names 'projection.*', 'point_net.*', 'mlp.*') into the packed
weight images the HIP kernels consume, once per set of weights, in fp64 on the host:

  * eval-mode BatchNorm is folded into the preceding conv / linear (source/base/nn.py:164-166,183-184,323-336,415);
  * fc1 (poco_model.py:368,405) is split: its latent part is hoisted from every (query, neighbour) pair to every
    POINT (`G = latents @ W1[:, :C]^T + b1`, pps_rows_dense256_f32), its xyz part stays per pair;
  * softmax weights sum to one, so fc_value/fc8 (poco_model.py:410-417) and the PointNet attention value conv
    (nn.py:89-93) commute with the pooling; together with the first MLP layer they compose into one 512->256
    layer evaluated once per query (pps_decode_tail_f32).

All of this is algebraically exact; results differ from the reference by fp32 re-association only
(tests/test_decoder_gpu.py, tolerance 1e-4 on the logits).
"""
import ctypes

import numpy as np
import torch

from . import _lib

BN_EPS = 1e-5
C = 256          # latent size the kernels are specialised for (configs/ppsurf.yaml:7)
HEADS = 64       # poco_model.py:374


def _np64(t):
    if isinstance(t, torch.Tensor):
        t = t.detach().cpu().numpy()
    return np.asarray(t, dtype=np.float64)


def _fold_bn(w, b, sd, bn):
    """(W [out,in], b [out]) followed by eval BatchNorm `bn` -> equivalent (W', b')."""
    scale = _np64(sd[bn + '.weight']) / np.sqrt(_np64(sd[bn + '.running_var']) + BN_EPS)
    return w * scale[:, None], (b - _np64(sd[bn + '.running_mean'])) * scale + _np64(sd[bn + '.bias'])


def _wb(sd, name):
    w = _np64(sd[name + '.weight'])
    return w.reshape(w.shape[0], -1), _np64(sd[name + '.bias'])


def pack_dense(w):
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty(_lib.lib().pps_packed_dense_floats(w.shape[0], w.shape[1]), dtype=np.float32)
    _lib.check(_lib.lib().pps_pack_dense_f32(w.ctypes.data, w.shape[0], w.shape[1], out.ctypes.data), 'pps_pack_dense_f32')
    return out


def pack_dense_f16x3(w):
    """[out,in] float -> uint16 image of the split-precision A operands (hi | lo f16 fragments, pps_pack_dense_f16x3)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty(_lib.lib().pps_packed_dense_f16x3_halfs(w.shape[0], w.shape[1]), dtype=np.uint16)
    _lib.check(_lib.lib().pps_pack_dense_f16x3(w.ctypes.data, w.shape[0], w.shape[1], out.ctypes.data), 'pps_pack_dense_f16x3')
    return out


def pack_xyz(w):
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty(_lib.lib().pps_packed_xyz_floats(w.shape[0]), dtype=np.float32)
    _lib.check(_lib.lib().pps_pack_xyz_f32(w.ctypes.data, w.shape[0], out.ctypes.data), 'pps_pack_xyz_f32')
    return out


def _pad(v, n):
    out = np.zeros(n, dtype=np.float32)
    out[:v.shape[0]] = v
    return out


class DecoderPlan:
    """Packed, device-resident weights of the eval-mode PPSurf decoder."""

    DTYPES = ('f32', 'f16x3')

    def __init__(self, sd, device, prefix='', dtype=None):
        """dtype 'f16x3' (default since round 3) or 'f32'; also through the environment variable PPS_DECODER_DTYPE or `network.decoder_dtype`.
        'f16x3': the dense layers of the interpolation branch (fc2, fc3, fc_query), of the PointNet branch (all but the xyz layers) and of the
        tail run on the f16 matrix pipe in split precision -- every fp32 product carried as three f16 MFMA products (hi.hi + hi.lo + lo.hi of
        x = hi + lo, ~21 significand bits) with fp32 accumulation (csrc/pps_common.h); the per-point table, the xyz layers and softmax / pooling
        stay fp32.  Held to the same 1e-4 bar as fp32 on the same cases (reference fixtures, oracle sweeps, full-size forward, config-5 chunk:
        tests/test_gpu_decoder.py, test_gpu_api.py, test_gpu_configs.py); wider than the fp16 autocast arithmetic the reference's own GPU
        predict runs in (configs/poco.yaml:10).  'f32': every product an fp32 MFMA, bit-for-bit an fp32 fmaf chain."""
        import os
        p = prefix
        self.dtype = dtype or os.environ.get('PPS_DECODER_DTYPE', 'f16x3')
        if self.dtype not in self.DTYPES:
            raise ValueError('decoder dtype must be one of {} (got {!r})'.format(self.DTYPES, self.dtype))
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        # ---- interpolation branch (poco_model.py:364-419) ------------------------------------------------
        w1, b1 = _wb(sd, p + 'projection.fc1')
        if w1.shape != (C, C + 3):
            raise NotImplementedError('the HIP decoder is specialised for latent size {} (got fc1 {})'.format(C, w1.shape))
        w2, b2 = _wb(sd, p + 'projection.fc2')
        w3, b3 = _wb(sd, p + 'projection.fc3')
        wq, bq = _wb(sd, p + 'projection.fc_query')
        wv, bv = _wb(sd, p + 'projection.fc_value')
        w8, b8 = _wb(sd, p + 'projection.fc8')
        if wq.shape[0] != HEADS or w8.shape[0] != C:
            raise NotImplementedError('unexpected projection head sizes {} {}'.format(wq.shape, w8.shape))
        host = {
            'g_w': pack_dense(w1[:, :C]), 'g_b': f32(b1),
            'ip_w': np.concatenate([pack_xyz(w1[:, C:]), pack_dense(w2), pack_dense(w3), pack_dense(wq)]),
            'ip_b': f32(np.concatenate([b2, b3, bq])),
        }
        # ---- PointNet branch (nn.py:255-373 with use_point_stn=False, use_feat_stn=True, sym_op='att') ---
        pn = p + 'point_net.'
        c0a = _fold_bn(*_wb(sd, pn + 'conv0a'), sd, pn + 'bn0a')
        c0b = _fold_bn(*_wb(sd, pn + 'conv0b'), sd, pn + 'bn0b')
        s1 = _fold_bn(*_wb(sd, pn + 'stn2.conv1'), sd, pn + 'stn2.bn1')
        s2 = _fold_bn(*_wb(sd, pn + 'stn2.conv2'), sd, pn + 'stn2.bn2')
        s3 = _fold_bn(*_wb(sd, pn + 'stn2.conv3'), sd, pn + 'stn2.bn3')
        f1 = _fold_bn(*_wb(sd, pn + 'stn2.fc1'), sd, pn + 'stn2.bn4')
        f2 = _fold_bn(*_wb(sd, pn + 'stn2.fc2'), sd, pn + 'stn2.bn5')
        f3w, f3b = _wb(sd, pn + 'stn2.fc3')
        if s3[0].shape != (256, 128) or f3w.shape != (4096, 64) or c0a[0].shape != (64, 3):
            raise NotImplementedError('the HIP PointNet is specialised for net_size_max=256, feature STN dim 64')
        c1 = _fold_bn(*_wb(sd, pn + 'conv1'), sd, pn + 'bn1')
        c2 = _fold_bn(*_wb(sd, pn + 'conv2'), sd, pn + 'bn2')
        c3 = _fold_bn(*_wb(sd, pn + 'conv3'), sd, pn + 'bn3')
        aq_w, aq_b = _wb(sd, pn + 'att.fc_query')
        av_w, av_b = _wb(sd, pn + 'att.fc_value')
        # conv1 (with its folded BatchNorm) acts on  trans2 @ x  (nn.py:327-334): W1 (T x) = (W1 T) x.  The last STN layer is linear in its input, so
        # it can emit M = W1 T directly: rows (o, b) of  f3m = sum_a W1[o, a] fc3[(a, b), :],  bias  W1 (mat(b3) + I).  The feature kernel then
        # applies ONE 64 x 64 matrix per query and conv1's bias + ReLU (24 of its 312 MFMAs per tile and one streamed weight chunk less)
        f3m = np.einsum('oa,abc->obc', c1[0], f3w.reshape(64, 64, 64)).reshape(4096, 64)
        f3mb = (c1[0] @ (f3b.reshape(64, 64) + np.eye(64))).reshape(-1)
        host['pa_w'] = np.concatenate([pack_xyz(c0a[0]), pack_dense(c0b[0]), pack_dense(s1[0]), pack_dense(s2[0]), pack_dense(s3[0])])
        host['pa_b'] = f32(np.concatenate([c0a[1], c0b[1], s1[1], s2[1], s3[1]]))
        host['pb_w'] = np.concatenate([pack_dense(f1[0]), pack_dense(f2[0]), pack_dense(f3m)])
        host['pb_b'] = f32(np.concatenate([f1[1], f2[1], f3mb]))                                # identity (nn.py:187-188) and conv1 inside
        # conv3 (+ bn3, no activation: nn.py:336) is followed by the attention pooling only, and both the attention logit and the value are
        # linear in its output: sum_p w_p Wv (W3 y_p + b3) = Wv W3 (sum_p w_p y_p) + Wv b3 (weights sum to one).  The feature kernel therefore pools
        # conv2's 128-channel output y and conv3 joins att.fc_value / the first MLP layer in the tail, evaluated once per QUERY instead of once per
        # patch point (192 of the 288 MFMAs of a tile, 144 of the 192 KB of streamed weights per tile)
        host['pc_w'] = np.concatenate([pack_xyz(c0a[0]), pack_dense(c0b[0]), pack_dense(c2[0])])
        # the attention logit of a patch point is linear in conv3's input: wq.(W3 y + b3) + bq = (W3^T wq).y + (wq.b3 + bq) (nn.py:88,336);
        # the kernels take u = W3^T wq and the constant, so the logit is known before conv3 runs (pps_decode.hip, feat_chain)
        u_att = c3[0].T @ aq_w.reshape(-1)
        s0_att = float(aq_w.reshape(-1) @ c3[1] + aq_b.reshape(-1)[0])
        host['pc_b'] = f32(np.concatenate([c0a[1], c0b[1], c1[1], c2[1], c3[1], u_att, np.zeros(128), [s0_att, 0.0, 0.0, 0.0]]))
        # ---- tail: fc_value . fc8 | att.fc_value, branch sum (ppsurf_model.py:100), MLP (nn.py:376-417) --
        m = p + 'mlp.layers.'
        l1 = _fold_bn(*_wb(sd, m + '0.0'), sd, m + '0.1')
        l2 = _fold_bn(*_wb(sd, m + '1.0'), sd, m + '1.1')
        l3w, l3b = _wb(sd, m + '2.0')
        if l3w.shape[0] != 2:
            raise NotImplementedError('occupancy head must have 2 outputs')
        wa = l1[0] @ w8 @ wv
        wb = l1[0] @ av_w @ c3[0]                                                               # 256 x 128: acts on the pooled conv2 output
        ba = l1[0] @ (w8 @ bv + b8 + av_w @ c3[1] + av_b) + l1[1]
        host['tl_w'] = np.concatenate([pack_dense(wa), pack_dense(wb), pack_dense(l2[0]), pack_dense(l3w)])
        host['tl_b'] = f32(np.concatenate([ba, l2[1], _pad(l3b, 32)]))
        expect = {'g_w': 65536, 'ip_w': 1024 + 65536 * 2 + 16384, 'ip_b': 576, 'pa_w': 256 + 4096 * 2 + 8192 + 32768,
                  'pa_b': 576, 'pb_w': 32768 + 8192 + 262144, 'pb_b': 128 + 64 + 4096, 'pc_w': 256 + 4096 + 8192,
                  'pc_b': 576 + 256 + 4, 'tl_w': 65536 + 32768 + 65536 + 8192, 'tl_b': 544}
        for k, n in expect.items():
            assert host[k].shape == (n,), (k, host[k].shape, n)
        self.device = torch.device(device)
        self.w = {k: torch.from_numpy(v).to(self.device) for k, v in host.items()}
        self.w16 = None
        if self.dtype == 'f16x3':
            split_layers = [w2, w3, wq, c0b[0], s1[0], s2[0], s3[0], f1[0], f2[0], f3m, c2[0], wa, wb, l2[0], l3w]
            wmax = max(float(np.abs(m_).max()) for m_ in split_layers)
            if not wmax < 65504.0:
                # a weight that f16 cannot hold: the split W = hi + lo does not exist.  (Activations are guarded on the device, see decode().)
                import warnings
                warnings.warn('decoder dtype f16x3 needs |weight| < 65504 (largest: {:.3g}); using the exact fp32 kernels'.format(wmax))
                self.dtype = 'f32'
        if self.dtype == 'f16x3':
            sets = [[w2, w3, wq], [c0b[0], s1[0], s2[0], s3[0]], [f1[0], f2[0], f3m], [c0b[0], c2[0]]]
            imgs = [np.concatenate([pack_dense_f16x3(m) for m in ms]) for ms in sets]
            # tail: the two parts of the (256 + 128) -> 256 layer alternate chunk by chunk (two output blocks each: 32 KiB of Wa, 16 KiB of Wb), then
            # L2, L3 (pps_decode_tail_f16x3)
            pa_, pb_ = pack_dense_f16x3(wa).reshape(8, -1), pack_dense_f16x3(wb).reshape(8, -1)
            ab = np.concatenate([np.concatenate([pa_[c], pb_[c]]) for c in range(8)])
            imgs.append(np.concatenate([ab, pack_dense_f16x3(l2[0]), pack_dense_f16x3(l3w)]))
            assert [i.shape[0] for i in imgs] == [2 * (65536 * 2 + 16384), 2 * 49152, 2 * (32768 + 8192 + 262144), 2 * (4096 + 8192),
                                                  2 * (65536 + 32768 + 65536 + 8192)]
            self._w16_t = [torch.from_numpy(i.view(np.int16)).to(self.device) for i in imgs]
            self.w16 = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in self._w16_t])
        self._scratch = {}

    # ---- scratch management: caller-owned buffers, reused across chunks -------------------------------------
    def scratch(self, name, shape, dtype=torch.float32):
        t = self._scratch.get(name)
        n = int(np.prod(shape))
        if t is None or t.numel() < n or t.dtype != dtype:
            old = t
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            t[:16].zero_()                                     # head of the decoder workspace: range-guard words (include/ppsurf_amd.h)
            if old is not None and old.dtype == dtype and old.numel() >= 16 and t.numel() >= 16:
                t[:16].copy_(old[:16])                         # the fall-back counter survives a growing workspace
            self._scratch[name] = t
        return t[:n].view(shape)

    def intermediates(self, q):
        """Views of the last decode() call's scratch (layout of pps_decode_fwd_f32): pooled [q,256], g [q,256],
        trans2 [q,4096] (= conv1 @ trans2 of the reference, 64 x 64 row-major: conv1 is folded into the STN's last layer), xbar [q,256] -- for
        tests and debugging."""
        ws = self.scratch('decode_ws', (_lib.lib().pps_decode_ws_bytes(q) // 4,))
        sizes = (('pooled', C), ('g', C), ('trans2', 4096), ('xbar', C))       # xbar: the first q x 128 floats of its slot hold the pooled conv2 output
        out, off = {}, 16                                      # 64 bytes of range-guard words first
        for name, width in sizes:
            out[name] = ws[off:off + q * width].view(q, width) if name != 'xbar' else ws[off:off + q * 128].view(q, 128)
            off += q * width
        return out

    # ---- kernels ---------------------------------------------------------------------------------------------
    def point_table(self, latents_cn):
        """G [N,256] = fc1_latent(latents) + b1.  `latents_cn` has SHAPE [C,N] like the reference's data['latents'][b]
        (source/ppsurf_model.py:78); a transposed view of point-major storage is read in place (strides decide)."""
        latents = latents_cn
        assert latents.dim() == 2 and latents.shape[0] == C and latents.dtype == torch.float32 and latents.device == self.device
        n, rs, cs = latents.shape[1], latents.stride(1), latents.stride(0)
        if cs == 1 and rs % 4 != 0:
            raise ValueError('point-major latents need a row stride that is a multiple of 4 floats')
        out = torch.empty((n, C), dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().pps_rows_dense256_f32(latents.data_ptr(), rs, cs, n, self.w['g_w'].data_ptr(),
                                                    self.w['g_b'].data_ptr(), out.data_ptr(), st), 'pps_rows_dense256_f32')
        return out

    def decode(self, table, pts, query, idx, patches, want_occ=True, stage_events=None, lane=0):
        """table G [N,256]; pts [N,3]; query [Q,3]; idx int64 [Q,k]; patches [Q,P,3] -> (logits [Q,2], occ [Q] | None): the whole
        chunk in one C call (pps_decode_fwd_f32).  stage_events: optional ctypes array of 6 hipEvent_t handles recorded around
        the five kernels of that same call (pps_decode_fwd_events_f32; bench.py).  lane: which scratch workspace the call uses -- calls that may
        run concurrently on different streams (ChunkPipeline lanes) must not share one."""
        L = _lib.lib()
        q, k, p = query.shape[0], idx.shape[1], patches.shape[1]
        for t in (table, pts, query, idx, patches):
            assert t.is_contiguous() and t.device == self.device
        st = torch.cuda.current_stream(self.device).cuda_stream
        logits = torch.empty((q, 2), dtype=torch.float32, device=self.device)
        occ = torch.empty((q,), dtype=torch.float32, device=self.device) if want_occ else None
        ws = self.scratch('decode_ws' if lane == 0 else 'decode_ws{}'.format(lane), (L.pps_decode_ws_bytes(q) // 4,))
        if getattr(self, '_wptrs', None) is None:
            self._wptrs = (ctypes.c_void_p * 10)(*[self.w[n].data_ptr() for n in ('ip_w', 'ip_b', 'pa_w', 'pa_b', 'pb_w', 'pb_b', 'pc_w',
                                                                                  'pc_b', 'tl_w', 'tl_b')])
        args = (table.data_ptr(), pts.data_ptr(), query.data_ptr(), idx.data_ptr(), q, k, patches.data_ptr(), p, self._wptrs,
                logits.data_ptr(), occ.data_ptr() if want_occ else None, ws.data_ptr())
        if self.w16 is not None:
            _lib.check(L.pps_decode_fwd_mixed_f32(*args[:9], self.w16, *args[9:], stage_events, st), 'pps_decode_fwd_mixed_f32')
        elif stage_events is None:
            _lib.check(L.pps_decode_fwd_f32(*args, st), 'pps_decode_fwd_f32')
        else:
            _lib.check(L.pps_decode_fwd_events_f32(*args, stage_events, st), 'pps_decode_fwd_events_f32')
        return logits, occ

    def range_fallbacks(self):
        """Number of chunks the split-precision path handed to the fp32 kernels so far because an activation left the f16 range (|x| > 65504);
        always 0 for dtype 'f32'.  Reads one device word (synchronises)."""
        if self.w16 is None:
            return 0
        return sum(int(ws[:16].view(torch.int32)[1]) for name, ws in self._scratch.items() if name.startswith('decode_ws'))


class ChunkPipeline:
    """Decodes a sequence of query chunks of ONE shape; replaces the chunk loops of source/poco_utils.py:218-223,146-153.

    lanes: consecutive chunks of a run() call are dealt to `lanes` HIP streams, each with its own neighbour tables, patches and decoder
    workspace.  A chunk's kernels are persistent grids that drain unevenly (the last tiles of the interpolation kernel leave most CUs idle, and
    the store-bound PointNet kernels leave the matrix pipe idle); with two lanes the next chunk's kernels fill those gaps: +5 % on a long chunk
    list (tools/time_chunk_overlap.py, profiles/NOTES_r5.md).  Results do not depend on the lane (chunks are independent, every lane runs the same
    kernels on its own buffers).  Default `lanes=None`: 2 for run() calls of at least LANE_MIN_CHUNKS chunks, else 1 -- a short list has nothing
    to overlap with and the second lane would only cost its scratch.
    overlap (single lane only): the spatial queries of chunk i+1 (kNN + patch gather: fp32 VALU) on a side stream underneath the decoder kernels of
    chunk i; double-buffered tables, events order the two streams."""

    LANE_MIN_CHUNKS = 4

    def __init__(self, plan: DecoderPlan, table, pts, raw, k: int, p: int, same_cloud: bool, max_chunk: int, overlap: bool = False, lanes=None):
        from . import ops
        import os
        self.ops, self.plan, self.table, self.pts, self.raw = ops, plan, table, pts, raw
        self.k, self.p, self.same_cloud, self.overlap = int(k), int(p), bool(same_cloud), bool(overlap)
        env = os.environ.get('PPS_CHUNK_LANES')
        self.lanes = int(env) if env else lanes                   # None: decided per run() call
        self.max_chunk = int(max_chunk)
        self.idx, self.patches = [], []
        self.pidx = None if (same_cloud and self.p <= self.k) else []
        self._grow_buffers(2)
        # the cloud is searched thousands of times per shape: arrange it once for the block-culling search
        self.blocks = self.ops.KnnBlocks(pts)
        self.raw_blocks = None if self.pidx is None else self.ops.KnnBlocks(raw)
        dev = plan.device
        self.side = torch.cuda.Stream(device=dev) if overlap else None
        self.lane_streams = []
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.free = [torch.cuda.Event() for _ in range(2)]
        self.n = 0

    def _grow_buffers(self, count):
        dev = self.plan.device
        while len(self.idx) < count:
            self.idx.append(torch.empty((self.max_chunk, self.k), dtype=torch.int64, device=dev))
            self.patches.append(torch.empty((self.max_chunk, self.p, 3), dtype=torch.float32, device=dev))
            if self.pidx is not None:
                self.pidx.append(torch.empty((self.max_chunk, self.p), dtype=torch.int64, device=dev))

    def _spatial(self, q, b):
        m = q.shape[0]
        idx = self.idx[b][:m]
        if self.pidx is not None and self.same_cloud and self.p > self.k:
            # one search serves both tables: the neighbours come back in ascending (distance, index) order, so the k nearest are the first k
            # columns of the P nearest (config 5: P = 200, k = 64; tests/test_gpu_knn.py checks the prefix property on tie-heavy clouds)
            src = self.pidx[b][:m]
            self.raw_blocks.query(q, self.p, out=src)
            idx.copy_(src[:, :self.k])
        else:
            self.blocks.query(q, self.k, out=idx)
            src = idx
            if self.pidx is not None:
                src = self.pidx[b][:m]
                self.raw_blocks.query(q, self.p, out=src)
        self.ops.patch_normalize(self.raw, q, src, self.p, out=self.patches[b][:m])

    def _run_lanes(self, chunks, want_occ, lanes):
        """Chunk i on lane i % lanes: every lane stream waits for the caller's stream once, runs its chunks in order on its own buffers, and the
        caller's stream waits for every lane at the end.  The result tensors are allocated here, on the caller's stream."""
        dev = self.plan.device
        main = torch.cuda.current_stream(dev)
        while len(self.lane_streams) < lanes:
            self.lane_streams.append(torch.cuda.Stream(device=dev))
        self._grow_buffers(max(lanes, 2))
        out = [None] * len(chunks)
        for ln in range(lanes):
            st = self.lane_streams[ln]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                for i in range(ln, len(chunks), lanes):
                    q = chunks[i]
                    m = q.shape[0]
                    self._spatial(q, ln)
                    out[i] = self.plan.decode(self.table, self.pts, q, self.idx[ln][:m], self.patches[ln][:m], want_occ=want_occ, lane=ln)
        for ln in range(lanes):
            main.wait_stream(self.lane_streams[ln])
        for lg, oc in out:                                      # allocated on a lane stream, consumed on the caller's: tell the allocator
            lg.record_stream(main)
            if oc is not None:
                oc.record_stream(main)
        return out

    def run(self, chunks, want_occ=True, stage_events=None):
        """chunks: list of contiguous float32 [q_i,3] device tensors -> list of (logits, occ).  stage_events (per-kernel HIP events, bench.py)
        forces a single lane: under two lanes a kernel's event time includes the moments it shares the chip."""
        lanes = self.lanes if self.lanes is not None else (2 if len(chunks) >= self.LANE_MIN_CHUNKS else 1)
        if stage_events is None and lanes > 1 and len(chunks) > 1 and self.side is None:
            res = self._run_lanes(chunks, want_occ, min(lanes, len(chunks)))
            self.n += len(chunks)
            return res
        main = torch.cuda.current_stream(self.plan.device)
        out = []
        for i, q in enumerate(chunks):
            b = (self.n + i) & 1
            if self.side is not None:
                self.side.wait_stream(main) if i == 0 else None
                with torch.cuda.stream(self.side):
                    self.side.wait_event(self.free[b])
                    self._spatial(q, b)
                    self.ready[b].record(self.side)
                main.wait_event(self.ready[b])
            else:
                self._spatial(q, b)
            m = q.shape[0]
            ev = stage_events[i] if stage_events is not None else None
            out.append(self.plan.decode(self.table, self.pts, q, self.idx[b][:m], self.patches[b][:m], want_occ=want_occ, stage_events=ev))
            self.free[b].record(main)
        self.n += len(chunks)
        return out


class PocoDecoderPlan:
    """Packed weights of POCO's projection head (source/poco_model.py:362-419; latent size 32 or 64, few output channels).
    fc1 is split like in DecoderPlan (per-point table G), fc8 . fc_value is composed on the host and applied after pooling."""

    DTYPES = ('f32', 'f16x3')

    def __init__(self, sd, device, prefix='projection', dtype=None):
        """dtype 'f16x3' or 'f32'; None = the environment variable PPS_DECODER_DTYPE, else 'f16x3' -- the same default and the same meaning as
        DecoderPlan: fc2, fc3 and fc_query carried as three f16 MFMA products per fp32 product with fp32 accumulation, behind the same device-side
        range guard (an activation beyond +-65504 makes the fp32 kernel queued behind recompute the call).  The layers are 32 or 64 wide and all
        weights sit in LDS either way; 'f16x3' is here so that one setting selects one arithmetic for both model classes."""
        import os
        self.dtype = dtype or os.environ.get('PPS_DECODER_DTYPE', 'f16x3')
        if self.dtype not in self.DTYPES:
            raise ValueError('decoder dtype must be one of {} (got {!r})'.format(self.DTYPES, self.dtype))
        p = prefix
        w1, b1 = _wb(sd, p + '.fc1')
        c = w1.shape[0]
        if c not in (32, 64) or w1.shape[1] != c + 3:
            raise NotImplementedError('the small-latent HIP decoder handles latent sizes 32 and 64 (got fc1 {})'.format(w1.shape))
        w2, b2 = _wb(sd, p + '.fc2')
        w3, b3 = _wb(sd, p + '.fc3')
        wq, bq = _wb(sd, p + '.fc_query')
        wv, bv = _wb(sd, p + '.fc_value')
        w8, b8 = _wb(sd, p + '.fc8')
        if wq.shape[0] != HEADS or w8.shape[0] > 8:
            raise NotImplementedError('unexpected projection head sizes {} {}'.format(wq.shape, w8.shape))
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self.c, self.nout, self.device = c, w8.shape[0], torch.device(device)
        self.g_w, self.g_b = f32(pack_dense(w1[:, :c])), f32(b1)
        self.w = f32(np.concatenate([pack_xyz(w1[:, c:]), pack_dense(w2), pack_dense(w3), pack_dense(wq)]))
        self.b = f32(np.concatenate([b2, b3, bq]))
        self.tail = f32(np.concatenate([(w8 @ wv).reshape(-1), w8 @ bv + b8]))
        self.w16 = None
        if self.dtype == 'f16x3':
            wmax = max(float(np.abs(m_).max()) for m_ in (w2, w3, wq))
            if not wmax < 65504.0:
                import warnings
                warnings.warn('decoder dtype f16x3 needs |weight| < 65504 (largest: {:.3g}); using the exact fp32 kernel'.format(wmax))
                self.dtype = 'f32'
        if self.dtype == 'f16x3':
            img = np.concatenate([pack_dense_f16x3(m_) for m_ in (w2, w3, wq)])
            assert img.shape[0] * 2 == (2 * c * c + HEADS * c) * 4                     # as many bytes as the fp32 packs it replaces in LDS
            self.w16 = torch.from_numpy(img.view(np.int16)).to(self.device)
            self._guard = torch.zeros(16, dtype=torch.int32, device=self.device)       # [0] range flag of the last call, [1] fall-back counter

    def point_table(self, latents_cn):
        """G [N,c] from latents of SHAPE [c,N] (any strides)."""
        lat = latents_cn.t().contiguous().float()
        out = torch.empty_like(lat)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().pps_rows_gemm_f32(lat.data_ptr(), None, self.c, None, None, 0, self.g_w.data_ptr(), self.g_b.data_ptr(), None, 0,
                                                lat.shape[0], self.c, out.data_ptr(), st), 'pps_rows_gemm_f32')
        return out

    def decode(self, table, pts, query, idx):
        """table [N,c]; pts [N,3]; query [Q,3]; idx int64 [Q,k] -> logits [Q,nout]."""
        q, k = query.shape[0], idx.shape[1]
        out = torch.empty((q, self.nout), dtype=torch.float32, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        if self.w16 is not None:
            _lib.check(_lib.lib().pps_interp_small_f16x3(table.data_ptr(), pts.data_ptr(), query.data_ptr(), idx.data_ptr(), q, k, self.c,
                                                         self.w.data_ptr(), self.w16.data_ptr(), self.b.data_ptr(), self.tail.data_ptr(), self.nout,
                                                         out.data_ptr(), self._guard.data_ptr(), st), 'pps_interp_small_f16x3')
            return out
        _lib.check(_lib.lib().pps_interp_small_f32(table.data_ptr(), pts.data_ptr(), query.data_ptr(), idx.data_ptr(), q, k, self.c,
                                                   self.w.data_ptr(), self.b.data_ptr(), self.tail.data_ptr(), self.nout, out.data_ptr(), st),
                   'pps_interp_small_f32')
        return out

    def range_fallbacks(self):
        """Calls the split-precision kernel handed to the fp32 kernel so far (an activation left the f16 range); 0 for dtype 'f32'.  Synchronises."""
        return 0 if self.w16 is None else int(self._guard[1])
