"""Prints ms_per_step of bench.py JSON lines matched by the glob arguments (mean and median): A/B runs of one box.

    python tools/show_bench_ms.py "gpurun_out/new_*.json" "gpurun_out/old_*.json"
"""
import json,sys,glob
for pat in sys.argv[1:]:
    vals=[]
    for f in sorted(glob.glob(pat)):
        try: vals.append(json.load(open(f))['ms_per_step'])
        except Exception as e: vals.append(None)
    v=[x for x in vals if x]
    print(pat, vals, 'mean %.3f median %.3f'%(sum(v)/len(v), sorted(v)[len(v)//2]) if v else '')
