"""PocoModel / PPSurfModel with the reference's constructor and Lightning hook signatures.

Mirrors source/poco_model.py:19-329 and source/ppsurf_model.py:10-36: `training_step(batch, batch_idx)`,
`validation_step`, `test_step`, `predict_step(batch, batch_idx, dataloader_idx=0)`, `compute_loss`, `calc_metrics`,
`self.network` with `.encoder/.projection/.point_net/.mlp` (state-dict names).  Works under pytorch_lightning when it is
installed and as a plain torch.nn.Module otherwise (ppsurf_amd.runner drives the hooks then).
"""
import math
import numbers
import os
import typing

import numpy as np
import torch
from torch import nn

from . import spatial, sharding
from .modules import _Base, PocoNetwork, PPSurfNetwork


def calc_accuracy(num_true, num_predictions):
    return float('NaN') if num_predictions == 0 else num_true / num_predictions


def calc_precision(tp, fp):
    return float('NaN') if tp + fp == 0 else tp / (tp + fp)


def calc_recall(tp, fn):
    return float('NaN') if tp + fn == 0 else tp / (tp + fn)


def calc_f1(precision, recall):
    if math.isnan(precision) or math.isnan(recall) or precision + recall == 0:
        return float('NaN')
    return 2.0 * (precision * recall) / (precision + recall)


def compare_predictions_binary_tensors(ground_truth, predicted, prediction_name):
    """source/base/metrics.py:41-84: confusion counts and accuracy / precision / recall / F1 of two {0,1} tensors."""
    if ground_truth.shape != predicted.shape:
        raise ValueError('The ground truth matrix and the predicted matrix have different sizes!')
    if not isinstance(ground_truth, torch.Tensor) or not isinstance(predicted, torch.Tensor):
        raise ValueError('Both matrices must be dense of type torch.tensor!')
    gt, pr = ground_truth > 0.0, predicted > 0.0
    res = {} if prediction_name is None else {'comp_name': prediction_name}
    n = float(gt.numel())
    tp, fp, fn = float((pr & gt).sum()), float((pr & ~gt).sum()), float((~pr & gt).sum())
    tn = n - tp - fp - fn
    res.update({'predictions': n, 'pred_gt': n, 'positives': tp + fp, 'pos_gt': tp + fn, 'true_neg': tn, 'negatives': n - tp - fp,
                'neg_gt': n - tp - fn, 'true_pos': tp, 'true': tp + tn, 'false_pos': fp, 'false_neg': fn, 'false': fp + fn})
    res['accuracy'] = calc_accuracy(res['true'], n)
    res['precision'] = calc_precision(tp, fp)
    res['recall'] = calc_recall(tp, fn)
    res['f1_score'] = calc_f1(res['precision'], res['recall'])
    return res


def binary_metrics_on_device(ground_truth, predicted):
    """accuracy / precision / recall / F1 of compare_predictions_binary_tensors as 0-dim DEVICE tensors (NaN where the
    reference returns NaN): no host synchronisation, for per-step logging inside the fit loop."""
    gt, pr = ground_truth > 0.0, predicted > 0.0
    n = float(gt.numel())
    tp, fp, fn = (pr & gt).sum().float(), (pr & ~gt).sum().float(), (~pr & gt).sum().float()
    nan = torch.full_like(tp, float('nan'))
    precision = torch.where(tp + fp == 0, nan, tp / (tp + fp))
    recall = torch.where(tp + fn == 0, nan, tp / (tp + fn))
    f1 = torch.where(torch.isnan(precision) | torch.isnan(recall) | (precision + recall == 0), nan, 2.0 * precision * recall / (precision + recall))
    accuracy = (n - fp - fn) / n if n > 0 else nan
    return {'accuracy': accuracy, 'precision': precision, 'recall': recall, 'f1_score': f1, 'abs_dist_rms': float('nan')}


def in_file_is_dataset(in_file: str):
    return os.path.splitext(in_file)[1].lower() == '.txt'


def get_results_dir(out_dir: str, name: str, in_file: str):
    return os.path.join(out_dir, name, os.path.basename(os.path.dirname(in_file)))


class PocoModel(_Base):

    def __init__(self, output_names, in_channels, out_channels, k, lambda_l1, debug, in_file, results_dir, padding_factor, name,
                 network_latent_size, gen_subsample_manifold_iter, gen_subsample_manifold, gen_resolution_global, rec_batch_size,
                 gen_refine_iter, workers):
        super().__init__()
        self.output_names, self.in_channels, self.out_channels, self.k = output_names, in_channels, out_channels, k
        self.lambda_l1, self.network_latent_size = lambda_l1, network_latent_size
        self.gen_subsample_manifold_iter, self.gen_subsample_manifold = gen_subsample_manifold_iter, gen_subsample_manifold
        self.gen_resolution_global, self.gen_resolution_metric, self.num_pts_local = gen_resolution_global, None, None
        self.rec_batch_size, self.gen_refine_iter, self.workers = rec_batch_size, gen_refine_iter, workers
        self.in_file, self.results_dir, self.padding_factor = in_file, results_dir, padding_factor
        self.debug, self.show_unused_params, self.name = debug, debug, name
        self.network = self._make_network()
        self.test_step_outputs = []
        self.last_prediction = None                 # (verts, faces) of the most recent predict_step, for callers/tests

    def _make_network(self):
        return PocoNetwork(in_channels=self.in_channels, latent_size=self.network_latent_size, out_channels=self.out_channels, k=self.k)

    # ---- progress bar / logging shims ---------------------------------------------------------------------------
    def get_prog_bar(self):
        trainer = self.__dict__.get('_runner_trainer')
        if trainer is None:
            try:
                trainer = self.trainer
            except Exception:
                trainer = None
        return getattr(trainer, 'progress_bar_callback', None) if trainer is not None else None

    def _log(self, *args, **kwargs):
        sink = self.__dict__.get('_fit_log')          # ppsurf_amd.fit collects the step's values here
        if sink is not None:
            return sink(*args, **kwargs)
        if hasattr(super(), 'log'):
            try:
                return super().log(*args, **kwargs)
            except Exception:
                return None
        return None

    def on_after_backward(self):
        if self.show_unused_params:
            for name, param in self.named_parameters():
                if param.grad is None:
                    print('Unused param {}'.format(name))
            self.show_unused_params = False

    # ---- loss / metrics (poco_model.py:75-118) ------------------------------------------------------------------
    def compute_loss(self, pred, batch_data):
        occ_loss = nn.functional.cross_entropy(input=pred, target=batch_data['occ'], reduction='none')
        loss_components = torch.stack([occ_loss])
        loss_components_mean = torch.stack([torch.mean(occ_loss)])
        return loss_components_mean.mean(), loss_components_mean, loss_components

    def calc_metrics(self, pred, gt_data):
        pred_labels = torch.argmax(pred, dim=1).to(torch.float32)
        eval_dict = compare_predictions_binary_tensors(ground_truth=gt_data['occ'].squeeze(), predicted=pred_labels.squeeze(),
                                                       prediction_name=None)
        eval_dict['abs_dist_rms'] = np.nan
        return eval_dict

    def get_loss_and_metrics(self, pred, batch):
        loss, mean, comps = self.compute_loss(pred=pred, batch_data=batch)
        return loss, mean, comps, self.calc_metrics(pred=pred, gt_data=batch)

    def default_step_dict(self, batch):
        pred = self.network.forward(batch)
        if self.__dict__.get('_fit_log') is not None and self.training:
            # inside ppsurf_amd.fit: keep the step free of host synchronisation (the reference's metric code calls .item()
            # several times between forward and backward); the values are read once, after the optimizer step is queued
            loss, mean, comps = self.compute_loss(pred=pred, batch_data=batch)
            metrics = binary_metrics_on_device(batch['occ'].squeeze(), torch.argmax(pred, dim=1).to(torch.float32).squeeze())
        else:
            loss, mean, comps, metrics = self.get_loss_and_metrics(pred, batch)
        if self.lambda_l1 != 0.0:
            raise NotImplementedError('lambda_l1 != 0 calls a regulariser that does not exist in the reference (poco_model.py:112-113)')
        return loss, mean, comps, metrics

    def training_step(self, batch, batch_idx):
        loss, mean, comps, metrics = self.default_step_dict(batch=batch)
        self.do_logging(loss, mean, log_type='train', output_names=self.output_names, metrics_dict=metrics, f1_in_prog_bar=False,
                        keys_to_log=frozenset({'accuracy', 'precision', 'recall', 'f1_score'}))
        return loss

    def validation_step(self, batch, batch_idx):
        loss, mean, comps, metrics = self.default_step_dict(batch=batch)
        self.do_logging(loss, mean, log_type='val', output_names=self.output_names, metrics_dict=metrics, f1_in_prog_bar=True,
                        keys_to_log=frozenset({'accuracy', 'precision', 'recall', 'f1_score'}))
        return loss

    def test_step(self, batch, batch_idx):
        pred = self.network.forward(batch)
        if batch['shape_id'].shape[0] != 1:
            raise NotImplementedError('batch size > 1 not supported')
        loss, mean, comps = self.compute_loss(pred=pred, batch_data=batch)
        metrics = self.calc_metrics(pred=pred, gt_data=batch)
        results = {'shape_id': batch['shape_id'].squeeze(0), 'pc_file_in': batch['pc_file_in'][0], 'loss': loss,
                   'loss_components_mean': mean.squeeze(0), 'loss_components': comps.squeeze(0), 'metrics_dict': metrics}
        self.test_step_outputs.append(results)
        bar = self.get_prog_bar()
        if bar is not None and getattr(bar, 'test_progress_bar', None) is not None:
            bar.test_progress_bar.set_postfix_str('pc_file: {}'.format(os.path.basename(results['pc_file_in'])), refresh=True)
        return results

    def on_test_epoch_end(self):
        """poco_model.py:164-181: per-shape table + means of the test metrics.  The reference writes an .xlsx through
        openpyxl (source/base/evaluation.py, not in the image); the same rows go to metrics_<name>.csv here."""
        from .data import read_shape_list
        if not self.test_step_outputs:
            return
        results_dir = get_results_dir(out_dir=self.results_dir, name=self.name, in_file=self.in_file)
        os.makedirs(results_dir, exist_ok=True)
        names = read_shape_list(self.in_file) if in_file_is_dataset(self.in_file) else [self.in_file]
        keys = ['accuracy', 'precision', 'recall', 'f1_score', 'abs_dist_rms']
        rows = []
        for out in self.test_step_outputs:
            sid = int(out['shape_id'])
            rows.append([names[sid] if sid < len(names) else str(sid), float(out['loss'])] + [float(out['metrics_dict'][k]) for k in keys])
        with open(os.path.join(results_dir, 'metrics_{}.csv'.format(self.name)), 'w') as f:
            f.write(','.join(['shape', 'loss'] + keys) + '\n')
            for r in rows:
                f.write(','.join([r[0]] + ['{:.6g}'.format(v) for v in r[1:]]) + '\n')
        arr = np.array([r[1:] for r in rows], dtype=np.float64)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', category=RuntimeWarning)             # abs_dist_rms is NaN for every shape
            mean = np.nanmean(arr, axis=0)
        print('Test results (mean): Loss={}, RMSE={}, F1={}'.format(mean[0], mean[5], mean[4]))
        self.test_step_outputs.clear()

    def on_predict_epoch_end(self):
        """poco_model.py:275-300: after reconstructing a DATASET the reference compares the meshes with ground truth
        (source/base/evaluation.py: Chamfer distance, IoU, normal error through trimesh/pysdf in worker processes).  That
        evaluation is outside the occupancy-query path; the guards are kept, the comparison is left to the reference's tools."""
        if not in_file_is_dataset(self.in_file):
            return
        gt_meshes_dir = os.path.join(os.path.dirname(self.in_file), '03_meshes')
        if not os.path.exists(gt_meshes_dir):
            print('Warning: {} not found. Skipping evaluation.'.format(gt_meshes_dir))
            return
        print('{}: meshes written to {}; quantitative comparison with {} is not part of ppsurf_amd (use the reference\'s '
              'source/base/evaluation.py on these files)'.format(self.name, get_results_dir(self.results_dir, self.name, self.in_file), gt_meshes_dir))

    def do_logging(self, loss_total, loss_components, log_type: str, output_names: list, metrics_dict: dict,
                   keys_to_log=frozenset({'abs_dist_rms', 'accuracy', 'precision', 'recall', 'f1_score'}), f1_in_prog_bar=True,
                   on_step=True, on_epoch=False):
        self._log('loss/{}/00_all'.format(log_type), loss_total, on_step=on_step, on_epoch=on_epoch)
        if len(loss_components) > 1:
            for li, l in enumerate(loss_components):
                self._log('loss/{}/{}_{}'.format(log_type, li, output_names[li]), l, on_step=on_step, on_epoch=on_epoch)
        for key, value in metrics_dict.items():
            if key in keys_to_log and torch.is_tensor(value):
                self._log('metrics/{}/{}'.format(log_type, key), torch.nan_to_num(value, nan=0.0), on_step=on_step, on_epoch=on_epoch)
            elif key in keys_to_log and isinstance(value, numbers.Number):
                self._log('metrics/{}/{}'.format(log_type, key), 0.0 if math.isnan(value) else value, on_step=on_step, on_epoch=on_epoch)
        self._log('metrics/{}/{}'.format(log_type, 'F1'), metrics_dict['f1_score'], on_step=on_step, on_epoch=on_epoch, logger=False,
                  prog_bar=f1_in_prog_bar)

    # ---- reconstruction (poco_model.py:183-273) -----------------------------------------------------------------
    @staticmethod
    def _draw_subset(covered, current_value, m, gen=None, device_rng=False):
        """The ids of one encoder pass (poco_model.py:210-224), or None when every point has been covered `current_value + 1`
        times.  All counts are >= current_value when round `current_value` runs, so the reference's loop condition
        `counts.min() < current_value + 1` (:209) is the same as "valid_ids is not empty" -- the nonzero() below is the only host
        synchronisation of a pass.  Random numbers as in the reference: the subset permutation comes from the CPU generator
        (`torch.randperm(valid_ids.shape[0])`, :213), the top-up permutation from the generator of the cloud's device (:217-219);
        `gen` (a seeded CPU generator shared by all ranks of a query-sharded run) replaces both.  device_rng: draw the subset
        permutation on the device as well -- same distribution, no host permutation of up to N elements + copy per pass; the
        subsets then no longer follow the reference's CPU random stream (PocoModel.latent_rng)."""
        n, dev = covered.shape[0], covered.device
        valid_ids = torch.nonzero(covered == current_value)[:, 0]
        if valid_ids.shape[0] == 0:
            return None
        if n < m:
            return torch.arange(n, device=dev)
        if device_rng and gen is None:
            ids = valid_ids[torch.randperm(valid_ids.shape[0], device=dev)[:m]]
        else:
            ids = valid_ids[torch.randperm(valid_ids.shape[0], generator=gen)[:m].to(dev)]
        if ids.shape[0] < m:
            # the reference draws the top-up with `torch.randperm(N, device=pts.device)` (:217-219): the generator of the cloud's device.  In
            # 'device' mode that is what happens here; in 'reference' mode the stream to follow is the one the reference's CPU execution
            # consumes (what the fixtures were recorded from), so the top-up comes from the CPU generator too
            if gen is not None:
                top = torch.randperm(n, generator=gen).to(dev)
            else:
                top = torch.randperm(n, device=dev) if device_rng else torch.randperm(n).to(dev)
            ids = torch.cat([ids, top[:m - ids.shape[0]]], dim=0)
        return ids

    @staticmethod
    def _draw_round(covered, current_value, m, gen=None):
        """ALL remaining subsets of coverage round `current_value` from ONE permutation: drawing m of the valid points, then m of the rest, ... is
        a uniformly random permutation of the valid points cut into consecutive pieces of m (the last piece topped up like _draw_subset does).
        Same distribution as pass-by-pass drawing with one nonzero() + one randperm() per ROUND instead of per pass (100 -> 10 host
        synchronisations and permutations per 100k-point cloud); not the reference's random stream, so only for the device / shared-generator
        streams.  [] when the round is complete."""
        n, dev = covered.shape[0], covered.device
        valid_ids = torch.nonzero(covered == current_value)[:, 0]
        v = valid_ids.shape[0]
        if v == 0:
            return []
        if n < m:
            return [torch.arange(n, device=dev)]
        perm = torch.randperm(v, device=dev) if gen is None else torch.randperm(v, generator=gen).to(dev)
        out = list(torch.split(valid_ids[perm], m))
        for piece in out:
            piece.pps_round = current_value               # full pieces of one round: no duplicate ids within or between them (encode_latents adds them at once)
        short = m - out[-1].shape[0]
        if short > 0:
            top = torch.randperm(n, device=dev) if gen is None else torch.randperm(n, generator=gen).to(dev)
            out[-1] = torch.cat([out[-1], top[:short]], dim=0)             # may repeat ids: a new, untagged tensor
        return out

    def _encode_subsets(self, pts_cf, subsets):
        """Latents of several equally sized subsets of one cloud in batched HIP launches: [B, m, C] point-major."""
        enc = self.network.encoder
        assert not enc.training
        if len(subsets) == 1:
            data_partial = {'pts': pts_cf[:, subsets[0]].unsqueeze(0)}
            data_partial.update(spatial.get_fkaconv_ids(data_partial))
            return enc.forward_point_major(data_partial, 0).unsqueeze(0)
        data_partial = {'pts': torch.stack([pts_cf[:, ids] for ids in subsets])}
        data_partial.update(spatial.get_fkaconv_ids(data_partial))
        return enc.forward_batch_point_major(data_partial)                     # folded BatchNorm, one launch sequence for all subsets

    @torch.no_grad()
    def encode_latents(self, pts_cf: torch.Tensor, progress=None, encode_subsets=None, trace=None) -> torch.Tensor:
        """Latent loop of poco_model.py:203-236 for one cloud.  pts_cf [3,N] on the device -> latents POINT-MAJOR [N,C]:
        coverage-balanced random subsets of gen_subsample_manifold points until every point has been encoded
        gen_subsample_manifold_iter times; latents are averaged.  Parity with the reference's loop (same torch seed -> same
        subsets, counts and latents, for `latent_batch` 1, 3 and 10): tests/test_driver_parity_cpu.py against
        tests/golden/latent_loop.npz.

        Which points a pass covers depends only on the coverage COUNTS, never on latents, so up to `latent_batch` (default 25)
        consecutive subsets -- across the boundaries of the coverage rounds -- are drawn exactly like the reference draws them one
        after the other and then encoded as ONE batch (batched sampling / kNN tables / FKAConv geometry and aggregation kernels,
        MFMA GEMMs with the folded BatchNorm over the rows of all subsets at once): a 10k-point pass alone cannot fill 256 CUs
        (round-3 probe time_latent_batch.py, git history: 106 / 93 / 89 / 91 ms per 100k-point cloud at 10 / 20 / 25 / 100 subsets per batch).

        With torch.distributed initialised and `shard_queries` set, the subsets of a batch are dealt round-robin to the ranks
        (the selection comes from a generator seeded identically on every rank) and the partial sums / counts are all-reduced
        once per batch.

        encode_subsets(pts_cf, [ids...]) -> [B,m,C] replaces the HIP encoder (tests); `trace` collects the ids of every pass."""
        n, dev = pts_cf.shape[1], pts_cf.device
        latent = torch.zeros((n, self.network_latent_size), dtype=torch.float32, device=dev)
        counts = torch.zeros((n,), dtype=torch.float32, device=dev)
        m = self.gen_subsample_manifold
        shard = bool(getattr(self, 'shard_queries', False)) and sharding.multi()      # several ranks (or one under PPS_SINGLE_RANK_COLLECTIVES)
        rank, world = sharding.world() if shard else (0, 1)
        gen = None
        if shard:
            gen = torch.Generator(device='cpu')
            gen.manual_seed(int(n) * 1000003 + 12345)
        encode = encode_subsets if encode_subsets is not None else self._encode_subsets
        batch = max(1, int(getattr(self, 'latent_batch', 25)))
        # 'reference': subsets follow torch's CPU generator like the reference (same seed -> same subsets; the parity tests);
        # 'device' (default on a GPU): the permutation is drawn where the counts live
        device_rng = getattr(self, 'latent_rng', 'device') == 'device' and dev.type == 'cuda'
        if shard:
            batch = -(-batch // world) * world                               # whole waves: every rank encodes batch / world subsets
        iteration = 0
        n_rounds = self.gen_subsample_manifold_iter
        # device / shared-generator streams: the subsets of a whole coverage round come from one permutation (_draw_round); the reference stream
        # is followed draw by draw
        per_round = batch > 1 and (device_rng or gen is not None)
        if getattr(self, 'latent_per_round', None) is not None:                  # tests: force either form
            per_round = bool(self.latent_per_round) and batch > 1

        def draw(covered, limit, state):
            """Up to `limit` further subsets exactly as the reference's loop draws them one after the other; `covered` is what its `counts` will hold
            by each draw (advanced here).  state = [current round, subsets of that round drawn ahead]."""
            out = []
            while len(out) < limit and state[0] < n_rounds:
                if per_round:
                    if not state[1]:
                        state[1].extend(self._draw_round(covered, state[0], m, gen))
                        if not state[1]:
                            state[0] += 1                                    # round complete: on with the next one, same counts
                            continue
                    ids = state[1].pop(0)
                else:
                    ids = self._draw_subset(covered, state[0], m, gen, device_rng)
                    if ids is None:
                        state[0] += 1
                        continue
                out.append(ids)
                if covered is not counts:
                    covered[ids] += 1                                        # what `counts[ids] += 1` will have done by the next draw
            return out

        state = [0, []]
        # batch > 1: EVERY subset is drawn before the first encoder pass.  Each draw synchronises with the device (nonzero: the number of valid
        # points); interleaved with the batches that wait drains the queue once per round and leaves the GPU idle while the host refills it.
        ahead = draw(counts.clone(), 1 << 30, state) if batch > 1 else None
        while True:
            if ahead is not None:
                subsets, ahead = ahead[:batch], ahead[batch:]
            else:
                subsets = draw(counts, 1, state)                             # pass by pass: counts are up to date, nothing is simulated
            if not subsets:
                break
            mine = subsets[rank::world] if shard else subsets
            part, cnt = (torch.zeros_like(latent), torch.zeros_like(counts)) if shard else (latent, counts)
            if mine:
                lat_b = encode(pts_cf, mine)
                i = 0
                while i < len(mine):
                    # consecutive full pieces of one coverage round share no id: one indexed add for all of them (the same single addition per
                    # element as subset by subset); anything else -- reference stream, topped-up pieces -- on its own
                    tag, j = getattr(mine[i], 'pps_round', None), i + 1
                    while tag is not None and j < len(mine) and getattr(mine[j], 'pps_round', None) == tag:
                        j += 1
                    ids = mine[i] if j == i + 1 else torch.cat(mine[i:j])
                    part[ids] += lat_b[i:j].reshape(ids.shape[0], -1).float()       # duplicate ids (top-up): one write wins, counted once, like the reference
                    cnt[ids] += 1
                    i = j
            if shard:
                sharding.allreduce_latents(part, cnt)
                latent += part
                counts += cnt
            if trace is not None:
                trace.extend(subsets)
            iteration += len(subsets)
            if progress is not None:
                progress('get_latent iter: {}'.format(iteration))
        return latent / counts.unsqueeze(1)

    @torch.no_grad()
    def predict_step(self, batch: dict, batch_idx, dataloader_idx=0):
        from . import reconstruct, meshio
        if batch['pts_ms'].shape[0] > 1:
            raise NotImplementedError('batch size > 1 not supported')
        self.network.eval()
        bar = self.get_prog_bar()
        progress = None
        if bar is not None and getattr(bar, 'predict_progress_bar', None) is not None:
            progress = lambda s: bar.predict_progress_bar.set_postfix_str(s, refresh=True)
        pc_file_in = batch['pc_file_in'][0]
        if in_file_is_dataset(self.in_file):
            out_file_rec = os.path.join(get_results_dir(self.results_dir, self.name, self.in_file), 'meshes', os.path.basename(pc_file_in))
        else:
            out_file_rec = os.path.join(self.results_dir, os.path.basename(pc_file_in), os.path.basename(pc_file_in) + '.ply')
        dev = next(self.network.parameters()).device
        pts_cf = torch.transpose(batch['pts_ms'], -1, -2)[0].to(dev).float()          # [3,N]   (get_data_poco, poco_data_loader.py:247)
        latent_pm = self.encode_latents(pts_cf, progress)
        shape = {'pts': pts_cf.unsqueeze(0), 'latents': latent_pm.t().unsqueeze(0)}     # [1,C,N] view of point-major storage
        mesh = reconstruct.export_mesh_and_refine_vertices_region_growing_v3(
            network=self.network, latent=shape, pts_raw_ms=batch['pts_raw_ms'] if 'pts_raw_ms' in batch else None,
            resolution=self.gen_resolution_global, padding=1, mc_value=0, num_pts=self.rec_batch_size, num_pts_local=self.num_pts_local,
            input_points=pts_cf.t().cpu().numpy(), refine_iter=self.gen_refine_iter, out_value=1, prog_bar=bar, pc_file_in=pc_file_in)
        self.last_prediction = mesh
        if getattr(self, 'shard_queries', False) and sharding.world()[0] != 0:
            return 0                                                   # every rank holds the same mesh; rank 0 writes it
        if mesh is not None:
            verts, faces = mesh
            if not in_file_is_dataset(self.in_file):               # de-normalise single files (poco_model.py:256-265)
                raw = meshio.load_pts(pc_file_in)[:, :3]
                bb_min, bb_max = raw.min(axis=0), raw.max(axis=0)
                verts = verts * (np.max(bb_max - bb_min) * (1.0 + self.padding_factor)) + (bb_min + bb_max) * 0.5
            meshio.write_ply_mesh(out_file_rec, verts, faces)
        else:
            print('No reconstruction for {}'.format(pc_file_in))
        return 0


class PPSurfModel(PocoModel):

    def __init__(self, pointnet_latent_size, output_names, in_channels, out_channels, k, lambda_l1, debug, in_file, results_dir,
                 padding_factor, name, network_latent_size, gen_subsample_manifold_iter, gen_subsample_manifold, gen_resolution_global,
                 num_pts_local, rec_batch_size, gen_refine_iter, workers):
        self._pps = (num_pts_local, pointnet_latent_size)
        super().__init__(output_names=output_names, in_channels=in_channels, out_channels=out_channels, k=k, lambda_l1=lambda_l1,
                         debug=debug, in_file=in_file, results_dir=results_dir, padding_factor=padding_factor, name=name,
                         workers=workers, rec_batch_size=rec_batch_size, gen_refine_iter=gen_refine_iter,
                         gen_subsample_manifold=gen_subsample_manifold, gen_resolution_global=gen_resolution_global,
                         gen_subsample_manifold_iter=gen_subsample_manifold_iter, network_latent_size=network_latent_size)
        self.num_pts_local, self.pointnet_latent_size = num_pts_local, pointnet_latent_size

    def _make_network(self):
        return PPSurfNetwork(in_channels=self.in_channels, latent_size=self.network_latent_size, out_channels=self.out_channels, k=self.k,
                             num_pts_local=self._pps[0], pointnet_latent_size=self._pps[1])
