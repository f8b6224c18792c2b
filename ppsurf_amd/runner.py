"""Minimal command-line runner with the reference's argument surface (pps.py:20-77, source/cli.py:43-118):

    python pps.py predict -c configs/poco.yaml -c configs/ppsurf.yaml -c configs/ppsurf_50nn.yaml \\
        --ckpt_path models/ppsurf_50nn/version_0/checkpoints/last.ckpt --trainer.devices 1 \\
        --data.init_args.in_file datasets/abc_minimal/testset.txt --model.init_args.gen_resolution_global 129
    python pps.py rec in_file.ply out_dir [overrides]            (pseudo-subcommand, pps.py:27-72)

Stacked `-c` YAML files (later overrides earlier), dotted overrides, `class_path` / `init_args` instantiation and the
argument links of poco.py:16-20 / pps.py:25.  Used when pytorch_lightning is not installed; with Lightning present the
reference's own LightningCLI can drive the same classes through the `source.*` import paths.
Subcommands: fit (ppsurf_amd/fit.py), test, predict.
"""
import copy
import importlib
import os
import sys

import torch
import yaml


class _Loader(yaml.SafeLoader):
    """YAML 1.2 floats: PyYAML (YAML 1.1) reads `eps: 1e-5` / `weight_decay: 1e-2` (configs/poco.yaml:66-67) as STRINGS; jsonargparse,
    which the reference's LightningCLI uses, coerces them to the annotated float type.  Resolve them as floats here."""


_Loader.add_implicit_resolver(
    'tag:yaml.org,2002:float',
    __import__('re').compile(r'^[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)$|^[-+]?(?:[0-9][0-9_]*)?\.[0-9_]+(?:[eE][-+]?[0-9]+)?$'),
    list('-+0123456789.'))


def _yaml_load(text):
    return yaml.load(text, Loader=_Loader)


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_dotted(cfg, dotted, value):
    keys = dotted.split('.')
    cur = cfg
    for k in keys[:-1]:
        cur = cur.setdefault(k, {})
    cur[keys[-1]] = _yaml_load(value)


def handle_rec_subcommand(args):
    """pps.py:27-72: `rec in_file out_dir [extra]` -> predict with the 50NN configs and checkpoint."""
    if len(args) <= 1 or args[1] != 'rec':
        return args
    if len(args) < 4:
        raise ValueError('Invalid syntax for rec subcommand: {}\nMake sure that it matches this example: '
                         'pps.py rec in_file.ply out_file.ply --model.init_args.rec_batch_size 50000'.format(' '.join(args)))
    in_file, out_dir = args[2], args[3]
    if not os.path.exists(in_file):
        raise ValueError('Input file does not exist: {}'.format(in_file))
    os.makedirs(out_dir, exist_ok=True)
    return args[:1] + ['predict', '-c', 'configs/poco.yaml', '-c', 'configs/ppsurf.yaml', '-c', 'configs/ppsurf_50nn.yaml',
                       '--ckpt_path', 'models/ppsurf_50nn/version_0/checkpoints/last.ckpt', '--data.init_args.in_file', in_file,
                       '--model.init_args.results_dir', out_dir, '--trainer.logger', 'False', '--trainer.devices', '1'] + args[4:]


def parse(argv):
    argv = handle_rec_subcommand(list(argv))
    if len(argv) < 2 or argv[1] not in ('fit', 'test', 'predict'):
        raise SystemExit('usage: pps.py {fit,test,predict,rec} [-c config.yaml ...] [--dotted.key value ...]')
    sub, cfg, ckpt, i = argv[1], {}, None, 2
    while i < len(argv):
        a = argv[i]
        if a in ('-c', '--config'):
            with open(argv[i + 1]) as f:
                _merge(cfg, _yaml_load(f) or {})
            i += 2
        elif a == '--ckpt_path':
            ckpt = argv[i + 1]
            i += 2
        elif a.startswith('--') and '=' in a:
            k, v = a[2:].split('=', 1)
            _set_dotted(cfg, k, v)
            i += 1
        elif a.startswith('--'):
            _set_dotted(cfg, a[2:], argv[i + 1])
            i += 2
        else:
            raise SystemExit('unexpected argument {!r}'.format(a))
    return sub, cfg, ckpt


def _link_arguments(cfg):
    """poco.py:16-20, pps.py:25."""
    m, d = cfg['model'].setdefault('init_args', {}), cfg['data'].setdefault('init_args', {})
    m['in_file'] = d.get('in_file')
    m['padding_factor'] = d.get('padding_factor')
    if 'num_pts_local' in m:
        d['num_pts_local'] = m['num_pts_local']
    return cfg


def _instantiate(spec):
    mod, cls = spec['class_path'].rsplit('.', 1)
    return getattr(importlib.import_module(mod), cls)(**spec.get('init_args', {}))


_DATA_CLASSES = {'source.poco_data_loader.PocoDataModule': 'ppsurf_amd.data.PocoDataModule',
                 'source.ppsurf_data_loader.PPSurfDataModule': 'ppsurf_amd.data.PPSurfDataModule'}


class _Bar:
    """stand-in for TQDMProgressBar: `predict_step` / `test_step` call set_postfix_str on these (poco_model.py:161,232)."""

    class _Line:
        def set_postfix_str(self, s, refresh=True):
            if os.environ.get('PPS_VERBOSE'):
                print(s, flush=True)

    predict_progress_bar = _Line()
    test_progress_bar = _Line()


class _Trainer:
    progress_bar_callback = _Bar()


def cpu_refusal(sub, argv, accel):
    """The one seam where this build differs from the reference's surface ("trainer.accelerator=cpu"): the reference's mini
    flow (configs/ppsurf_mini.yaml, README "--trainer.accelerator cpu", pps.py:27-72) runs its PyTorch modules on the host; here every op of the
    path is a HIP kernel and a CPU twin inside the product would be a second, unverified implementation.  The message names the command to run
    instead: the same one on the GPU."""
    args, skip = [], False
    for a in argv[1:]:
        if skip:
            skip = False
            continue
        if a == '--trainer.accelerator':
            skip = True
            continue
        if a.startswith('--trainer.accelerator='):
            continue
        args.append(a)
    cmd = 'python pps.py ' + ' '.join(args + ['--trainer.accelerator', 'gpu', '--trainer.devices', '1'])
    why = 'trainer.accelerator={}'.format(accel) if accel == 'cpu' else 'no GPU is visible (torch.cuda.is_available() is False)'
    return ('ppsurf_amd `{}` refused: {}.  This build has no CPU path -- every operator of the occupancy path is a HIP kernel for MI355X (gfx950) '
            'and the product never falls back to a host implementation (the CPU restatement under oracle/ is test infrastructure only).  '
            'Run the same configuration on the GPU instead:\n    {}\n(the plumbing-only flow of configs/ppsurf_mini.yaml is covered in that form '
            'by tests/test_gpu_configs.py::test_config1_*; see INTEGRATION.md, "trainer.accelerator=cpu")'.format(sub, why, cmd))


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    sub, cfg, ckpt = parse(argv)
    if cfg.get('seed_everything') is not None:
        import random
        import numpy as np
        seed = int(cfg['seed_everything'])
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    cfg = _link_arguments(cfg)
    accel = str(cfg.get('trainer', {}).get('accelerator', 'gpu'))
    if accel == 'cpu' or not torch.cuda.is_available():
        raise RuntimeError(cpu_refusal(sub, argv, accel))
    # one process per GPU; more ranks than GPUs (2-rank rehearsals on a 1-GPU box) wrap around
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    from . import sharding
    if world > 1 or sharding.single_rank_collectives():
        torch.cuda.set_device(device)
        sharding.init_process_group(device)                     # PPS_BACKEND (default nccl = RCCL), 127.0.0.1
    model = _instantiate(cfg['model']).to(device).eval()
    dspec = dict(cfg['data'])
    dspec['class_path'] = _DATA_CLASSES.get(dspec['class_path'], dspec['class_path'])
    data = _instantiate(dspec)
    data.device = device
    model.__dict__['_runner_trainer'] = _Trainer()
    if sub == 'fit':
        # data-parallel over shapes: every rank runs its own batches, gradients are averaged over RCCL (ppsurf_amd/fit.py);
        # data.init_args.use_ddp selects the DistributedSampler semantics like in the reference
        from . import fit as fit_mod
        fit_mod.fit(model, data, cfg, ckpt_path=ckpt, device=device)
        return model
    if ckpt is not None:
        state = torch.load(ckpt, map_location='cpu')
        model.load_state_dict(state.get('state_dict', state))
    # multi-GPU predict (launched with torch.distributed.run, one rank per GPU, RCCL): PPS_SHARD=shapes (default) deals the
    # shapes of the test set round-robin to the ranks, no communication; PPS_SHARD=queries shards the query blocks and the
    # encoder passes of every shape over all ranks (latent all-reduce + per-round all-gather of occupancies).
    shard = os.environ.get('PPS_SHARD', 'shapes')
    if sharding.multi():
        model.shard_queries = shard == 'queries'
        sharding.set_query_sharding(model.shard_queries)
    with torch.no_grad():
        if sub == 'predict':
            from .fit import HostGcPacer
            with HostGcPacer(every=1) as pacer:                 # a young-generation collection between shapes, none inside one (fit.HostGcPacer)
                for i, batch in enumerate(data.predict_dataloader()):
                    if world > 1 and shard == 'shapes' and i % world != rank:
                        continue
                    model.predict_step(batch, i)
                    pacer.tick()
            model.on_predict_epoch_end()
        else:
            for i, batch in enumerate(data.test_dataloader()):
                out = model.test_step(batch, i)
                print('{}: loss {:.6f} f1 {:.4f}'.format(os.path.basename(out['pc_file_in']), float(out['loss']), out['metrics_dict']['f1_score']))
            model.on_test_epoch_end()
    return model


if __name__ == '__main__':
    main()
