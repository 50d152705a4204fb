import json, sys
t = sys.stdin.read(); d = json.loads(t[t.index("{"):])
print({k: v for k, v in d.items() if k != "layers" and "library" not in k})
print(" ".join("%s:%sus/%s" % (k.split("_")[0], v["us"], v["TFLOPs"]) for k, v in d["layers"].items()))
