import importlib, sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
P = importlib.import_module("small-object-detection-transformers_amd.preprocess")
g = torch.Generator().manual_seed(2*1000+64+2)
rgb = torch.randint(0,256,(2,3,64,64),generator=g,dtype=torch.uint8)
ir = torch.randint(0,256,(2,3,64,64),generator=g,dtype=torch.uint8)
o1,o2 = P.preprocess_batch(rgb.cuda(), ir.cuda(), 2)
r1 = F.interpolate(rgb.float()/255.0, size=[32,32], mode='bilinear', align_corners=True)
d = (o1.cpu()-r1).abs()
print(d.max(), (d>1e-6).sum(), (d>1e-6).nonzero()[:10])
i = d.argmax(); b,c,y,x = [int(v) for v in torch.unravel_index(i, d.shape)]
print(b,c,y,x, float(o1[b,c,y,x]), float(r1[b,c,y,x]))
sy = 63.0/31.0
print("src y", y*sy, "x", x*sy, rgb[b,c,int(y*sy):int(y*sy)+2, int(x*sy):int(x*sy)+2])
