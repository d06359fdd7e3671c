import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if k=='summary': continue
    print(f"{k:34s}", *(f"{p}: worst {v[p]['worst_rel_l2']:.2e} w {v[p]['weights_worst']:.2e} b {v[p]['bias_worst']:.2e} med {v[p]['median_rel_l2']:.2e} |" for p in v))
