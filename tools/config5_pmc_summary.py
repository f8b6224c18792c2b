"""gpurun_out/prof_c5_<tag>/summary_pmc.json (tools/rocpd_summary.py) -> config5_pmc.json: HBM bytes per launch of the k = 200 block search and of the
patch kernel of `python bench.py --only config5` (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes: MI355X_MICROARCH.md, HBM section), tied to the
kernel sources by their digest.     python tools/config5_pmc_summary.py gpurun_out/prof_c5_r06 out.json"""
import json
import os
import sys


def main(src, dst):
    d = json.load(open(os.path.join(src, 'summary_pmc.json')))
    k = d['kernels']

    def pick(prefix):
        # the search kernel of the timed region: the one with the most dispatches among the names that match
        best = None
        for name, v in k.items():
            if name.startswith(prefix) and 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
                if best is None or v.get('calls', 0) > best[1].get('calls', 0):
                    best = (name, v)
        return best

    out = {'source': d['source'], 'git_head': d['git_head'], 'csrc_digest': d['csrc_digest'], 'command': 'PPS_CHUNK_LANES=1 python bench.py --only config5'}
    for key, prefix in (('knn', 'knn_blocked'), ('patch', 'patch_normalize')):
        hit = pick(prefix)
        if hit is None:
            continue
        name, v = hit
        out[key + '_kernel'] = name
        out[key + '_hbm_bytes_per_launch'] = (2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0
        out[key + '_fetch_kib'], out[key + '_write_kib'] = v['FETCH_SIZE'], v['WRITE_SIZE']
        out[key + '_avg_us'] = v.get('avg_us')
        out[key + '_calls'] = v.get('calls')
    json.dump(out, open(dst, 'w'), indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
