import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d["stages"]
print(sys.argv[1].split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], "map_eq_s", d["breakdown"]["map_eq_s"], "tail", d["breakdown"]["tail_s(eq_export+normalize+EM)"])
print("   ", " ".join("%s=%.3f" % (k, v["avg_ms"]) for k, v in s.items()))
