"""wall time of the module-API paths at 256^3 (configs[2] pieces): stage-1, stage-2 (2E3) and the packers, fwd and fwd+bwd."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsenet_amd as H
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
img, patch = (256, 256, 256), (16, 16, 16)
kw = dict(in_channels=1, img_size=img, patch_size=patch, pos_embed="perceptron", classification=True)
s1, s2 = H.ViT_stage1(**kw).to(dev), H.ViT_stage2(**kw).to(dev)
pk = H.VisualPacker_3d_phi_v3(img, patch, 768, 3072, "mlp", 2, grid=(16, 16, 16), kernel_size=(2, 4, 4)).to(dev)
vol = torch.rand(B, 1, *img, device=dev)
i2d = torch.randn(B, 32, 768, device=dev)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
def f1():
    with torch.no_grad(): s1(vol)
def f2():
    with torch.no_grad(): s2(vol, i2d)
def fb1():
    y, _ = s1(vol); (y * y).mean().backward()
def fb2():
    y, _ = s2(vol, i2d); (y * y).mean().backward()
feat = torch.randn(B, 4096, 768, device=dev, requires_grad=True)
def fp():
    with torch.no_grad(): pk(feat)
def fbp():
    y = pk(feat); (y * y).mean().backward()
print(f"B={B}: stage1 fwd {t(f1):.1f} ms | stage2 fwd {t(f2):.1f} ms | stage1 f+b {t(fb1):.1f} ms | stage2 f+b {t(fb2):.1f} ms | packer fwd {t(fp):.2f} ms | packer f+b {t(fbp):.2f} ms")
