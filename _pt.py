import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import flex_amd, torch
torch.cuda.init()
a = flex_amd.synth_graph(sys.argv[1])
for i in range(3):
    t0 = time.time()
    p = flex_amd.Plan(a, 128, order=flex_amd.FLEX_ORDER_CLUSTER)
    print("plan s", time.time() - t0, flush=True)
    del p
