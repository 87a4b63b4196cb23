import sys, time, torch
sys.path.insert(0, '.')
from incomplete_multimodal_fusion_amd.engine import FlatAdamW
from incomplete_multimodal_fusion_amd.pretrain import PretrainStep, get_model
import torch.cuda.tunable as tun, shutil, os
shutil.copyfile('incomplete_multimodal_fusion_amd/tuned/tunableop_gfx950.csv', '/tmp/t.csv')
tun.enable(True); tun.tuning_enable(True); tun.set_filename('/tmp/t.csv'); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30)
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(1)
B = 256
x = {"s1": torch.randn(B, 1, 256, 256, generator=g).to(dev), "s2": torch.randn(B, 3, 256, 256, generator=g).to(dev), "dem": torch.randn(B, 1, 256, 256, generator=g).to(dev)}
for outs in (('s1', 's2', 'dem'), ('s1',)):
    torch.manual_seed(0)
    model = get_model('base', input_size=256, out_domains=outs).to(dev).train()
    opt = FlatAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.05, exclude=model.never_used_parameters())
    step = PretrainStep(model, opt, 384, autocast=True)
    for _ in range(4): step(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): step(x)
    torch.cuda.synchronize(); print(outs, "%.2f ms/step" % ((time.perf_counter() - t0) / 8 * 1e3), flush=True)
    del model, opt, step
