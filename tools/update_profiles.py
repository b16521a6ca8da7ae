#!/usr/bin/env python3
"""Copy the summaries of one tools/profile_round.sh run into profiles/ and recompute profiles/traffic.json.
   python tools/update_profiles.py gpurun_out/prof_<tag> r02 <commit>"""
import json
import os
import re
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, 'profiles')


def json_line(path):
    return [l for l in open(path) if l.startswith('{')][-1]


open(os.path.join(dst, f'{tag}_bench.json'), 'w').write(json_line(os.path.join(src, 'bench.json')))
open(os.path.join(dst, f'{tag}_bench_under_rocprof.json'), 'w').write(json_line(os.path.join(src, 'bench_under_rocprof.json')))
for a, b in (('kernel_stats.csv', 'bench_kernel_stats.csv'), ('kernel_durations.txt', 'kernel_durations.txt'),
             ('pmc_fetch_size.txt', 'pmc_fetch_size.txt'), ('pmc_write_size.txt', 'pmc_write_size.txt')):
    shutil.copy(os.path.join(src, a), os.path.join(dst, f'{tag}_{b}'))


def parse(p):
    out, name = {}, None
    for l in open(p):
        if not l[0].isspace():
            name = re.search(r'spmm_(?:stage|bundle|hot|dense|strip)_kernel(?:<[^>]*>)?', l).group(0)
        else:
            m = re.search(r'n=\s*(\d+) mean=([\d.e+]+) sum=([\d.e+]+)', l)
            if m:
                out[name] = (int(m.group(1)), float(m.group(3)))
    return out


f = parse(os.path.join(dst, f'{tag}_pmc_fetch_size.txt'))
w = parse(os.path.join(dst, f'{tag}_pmc_write_size.txt'))
n = sum(v[0] for v in f.values())
fetch = sum(v[1] for v in f.values()) * 1024 * 2          # KiB; doubled on gfx950 (MI355X_MICROARCH.md, HBM section)
write = sum(v[1] for v in w.values()) * 1024
per = {k: dict(launches=f[k][0], fetch_bytes_per_launch=int(f[k][1] * 2048 / f[k][0]),
               write_bytes_per_launch=int(w[k][1] * 1024 / w[k][0])) for k in f}
d = dict(hbm_bytes_per_launch=int((fetch + write) / n), launches_in_pass=n, fetch_bytes_per_launch=int(fetch / n),
         write_bytes_per_launch=int(write / n), per_kernel=per, commit=(sys.argv[3] if len(sys.argv) > 3 else 'unknown'),
         command='DGNN_TUNE_GEMM=0 python bench.py --no-alt --cpu-budget 0 --steps 3 --warmup 2',
         how="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `DGNN_TUNE_GEMM=0 python bench.py --cpu-budget 0 "
             "--steps 3 --warmup 2` (library-default GEMMs there; the SpMM launches are the same) (KiB per dispatch; FETCH_SIZE doubled for gfx950 per MI355X_MICROARCH.md), averaged "
             "over all SpMM stage launches of the pass (spmm_strip_kernel, spmm_dense_kernel, spmm_bundle_kernel and spmm_stage_kernel, whichever ran); counts L2-side fabric requests, Infinity-Cache hits included")
json.dump(d, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
print(json.dumps(d, indent=1))
