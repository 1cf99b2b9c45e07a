import sys, json
for l in sys.stdin:
    if not l.startswith("{"): continue
    d = json.loads(l); pc = d["phase_cycles"]; tot = sum(pc) or 1
    print("ms %.2f iters %.2fM bal %.3f | gen %.1f%% trace %.1f%% shade %.1f%% mesh %.1f%% (wave-cycles %.3g)" % (min(d["kernel_ms"]), d["wave_iterations"]/1e6, d["balance"], 100*pc[0]/tot, 100*pc[1]/tot, 100*pc[2]/tot, 100*pc[3]/tot, tot))
