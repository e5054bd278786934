import re,sys
for f in sys.argv[1:]:
    n=0; bg=lg=sm=0.0; nl=0; cb=th=0.0
    for line in open(f, errors="replace"):
        m=re.search(r"sumcheck_layer: nterms (\d+) nh0 (\d+) nw (\d+) logw (\d+) \| bind_g (\d+) us \| (\d+) large round-hands (\d+) us \| (\d+) small (\d+) us \| caller's round callback ([\d.]+) us in all, host between post and answer ([\d.]+)", line)
        if not m: continue
        n+=1; bg+=float(m.group(5)); nl+=int(m.group(6)); lg+=float(m.group(7)); sm+=float(m.group(9)); cb+=float(m.group(10)); th+=float(m.group(11))
    p=n/13.0
    print(f, "layers",n,"proofs %.0f"%p, "per proof: bind_g %.2f ms, large %.1f round-hands %.2f ms, small %.2f ms, callback %.2f ms, think %.2f ms"%(bg/p/1e3, nl/p, lg/p/1e3, sm/p/1e3, cb/p/1e3, th/p/1e3))
