import json, sys
d0 = sys.argv[1]
for x in "abcfwd":
    try:
        d = json.load(open(f"{d0}/{x}.json"))
        for k in d:
            if "dr_hl" in k["kernel"] or "80, 128, 32, 1, 4, 2" in k["kernel"] or "dr_tn" in k["kernel"]:
                print(x, {a: (round(b) if isinstance(b, float) else b) for a, b in k.items()})
    except Exception as e:
        print(x, "ERR", e)
