import sys,json
d=json.load(sys.stdin); r=d["roofline"]
print(sys.argv[1], d["config"]["integrator_form"][:14], "%.2f us steady %.2f us frac %.3f" % (r["launch_ms"]*1e3, r["launch_ms_steady"]*1e3, r["frac_steady"]))
