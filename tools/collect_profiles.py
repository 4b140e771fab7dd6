"""python tools/collect_profiles.py <round tag, e.g. r02> "<what was profiled>"
Copies the summaries of the last tools/profile_round.sh run (gpurun_out/prof_round/) into profiles/ under the round's name
and derives profiles/<tag>_traffic.json (HBM bytes per launch per kernel) from the PMC summary."""
import glob, json, os, shutil, sys
tag, what = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, 'gpurun_out', 'prof_round'), os.path.join(R, 'profiles')
shutil.copy(glob.glob(os.path.join(O, 'graph', '**', 'g_kernel_stats.csv'), recursive=True)[0], os.path.join(P, f'{tag}_kernel_stats_hipgraph_1000rays.csv'))
shutil.copy(glob.glob(os.path.join(O, 'eager', '**', 'e_kernel_stats.csv'), recursive=True)[0], os.path.join(P, f'{tag}_kernel_stats_eager_1000rays.csv'))
shutil.copy(os.path.join(O, 'timeline.txt'), os.path.join(P, f'{tag}_timeline_one_replay.txt'))
pmc = json.load(open(os.path.join(O, 'pmc_summary.json')))
json.dump(pmc, open(os.path.join(P, f'{tag}_pmc_summary.json'), 'w'), indent=1)
out = {"_how": "tools/profile_round.sh on the GPU box: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum TCC_MISS_sum "
               "(separate passes) -- python3 bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline --no-kernel-events --no-api "
               "(config 2 only, the HEADLINE mode: hipGraph replays unless PMC_MODE=--eager was set); mean per dispatch; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads half of wide "
               f"coalesced fetches, MI355X_MICROARCH.md; WRITE_SIZE exact, float atomics included). Full counter set: {tag}_pmc_summary.json",
       "_commit": what}
for k, v in pmc.items():
    if not isinstance(v, dict) or 'FETCH_SIZE' not in v:
        continue
    name = k.split('<')[0]
    out[name] = {"FETCH_SIZE_KB": v['FETCH_SIZE'], "WRITE_SIZE_KB": v['WRITE_SIZE'],
                 "bytes_per_launch": int((2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024),
                 "TCC_HIT_sum": v.get('TCC_HIT_sum'), "TCC_MISS_sum": v.get('TCC_MISS_sum'), "dispatches": v.get('dispatches')}
mall = os.path.join(O, 'mall.txt')
out["_infinity_cache"] = open(mall).read().strip() if os.path.exists(mall) else "not probed"
json.dump(out, open(os.path.join(P, f'{tag}_traffic.json'), 'w'), indent=1)
print(open(os.path.join(P, f'{tag}_timeline_one_replay.txt')).read())
for k in out:
    if not k.startswith('_'):
        print(k, out[k]['bytes_per_launch'])
print(out["_infinity_cache"])
