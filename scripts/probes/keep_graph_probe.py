import ctypes, torch
hiprt = ctypes.CDLL("libamdhip64.so")
def count_nodes(g):
    raw = g.raw_cuda_graph()
    n = ctypes.c_size_t(0)
    rc = hiprt.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n))
    nodes = (ctypes.c_void_p * n.value)()
    rc2 = hiprt.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n))
    types = []
    for i in range(n.value):
        t = ctypes.c_int(-1)
        hiprt.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        types.append(t.value)
    return rc, rc2, n.value, types
x = torch.ones(1000, device="cuda"); y = torch.zeros(1000, device="cuda")
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    y.add_(x); y.mul_(2.0); y.copy_(x)
print("nodes", count_nodes(g))
try:
    g.replay(); torch.cuda.synchronize(); print("replay without instantiate ok", y[0].item())
except Exception as e:
    print("replay failed:", e)
    g.instantiate(); g.replay(); torch.cuda.synchronize(); print("after instantiate", y[0].item())
