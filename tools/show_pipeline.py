import json,sys,numpy as np
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=d["runs"][3:]
    print(f.split('/')[-1], "verify ms median %.4f"%np.median([x["t_verify_ms"] for x in r]), "cands", int(np.median([x["candidates"] for x in r])), "congruent %.3f"%np.median([x["t_congruent_ms"] for x in r]), "transforms %.3f"%np.median([x["t_transforms_ms"] for x in r]), "sample %.3f"%np.median([x["t_sample_ms"] for x in r]), "best", [round(x["best_lcp"],4) for x in r[:4]])
