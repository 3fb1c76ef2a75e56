import glob, json, sys
for f in sorted(glob.glob('gpurun_out/sweep_*.json')):
    try:
        d = json.loads(open(f).read())
        k = d['roofline']['all_kernels_ms']
        print(f"{f:38s} {d['value']:9.1f} env-steps/s {d['ms_per_step']:8.3f} ms/step  E={d['config']['envs_per_gpu']}")
        print("     " + "  ".join(f"{n.replace('_kernel','')}={v*1e3:.1f}us" for n, v in k.items()))
        for n, h in d['roofline'].get('hbm_kernels', {}).items():
            print(f"     {n}: {h['achieved']} GB/s ({h['frac']*100:.1f}% of 8 TB/s)")
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-400:])
