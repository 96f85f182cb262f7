import os, sys
sys.path.insert(0, os.getcwd())
import poseidon_amd as pa
nm, npods = int(sys.argv[1]), int(sys.argv[2])
cl = pa.synth_cluster(0, 7, nm, npods)
with pa.Solver() as s:
    net = s.build_network(cl)
    for _ in range(2):
        s.eval_costs(cl, net)
        r = s.solve()
