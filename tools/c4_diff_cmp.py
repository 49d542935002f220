import json,sys,glob
base=json.load(open("/tmp/d_serial.json"))
for f in sorted(glob.glob("/tmp/d_*.json")):
    d=json.load(open(f))
    nd=sum(1 for a,b in zip(base["rows"],d["rows"]) if a!=b); ids=sorted({a[0] for a,b in zip(base["rows"],d["rows"]) if a!=b})
    print(f, "rows differing from serial:", nd, ids, "selfcheck differs:", [k for k in base["dev_f"] if base["dev_f"][k]!=d["dev_f"][k]])
