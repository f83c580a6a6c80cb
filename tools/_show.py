import json,sys,glob
for pat in sys.argv[1:]:
    vals=[]
    for f in sorted(glob.glob(pat)):
        try: vals.append(json.load(open(f))['ms_per_step'])
        except Exception as e: vals.append(None)
    v=[x for x in vals if x]
    print(pat, vals, 'mean %.3f median %.3f'%(sum(v)/len(v), sorted(v)[len(v)//2]) if v else '')
