import sys, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_golden, t
from flowmap_amd.loss.mapping import get_mapping
from helpers import mapping_cfg
g = load_golden("fn_mapping")
dev = "cuda:0"
a = t(g["a"]).to(dev).requires_grad_(True)
val = get_mapping(mapping_cfg("l1")).forward(a, t(g["b"]).to(dev), tuple(int(x) for x in g["image_shape"]))
val.sum().backward()
d = (a.grad.cpu() - t(g["l1_g_a"])).abs()
print(d.max(), d.argmax(), a.grad[:3].cpu(), t(g["l1_g_a"])[:3], val[:3].cpu(), t(g["l1_val"])[:3])
print(t(g["a"])[:2], t(g["b"])[:2])
