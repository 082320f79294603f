"""swin_T / swin_S (embed 96 -> 192-wide StyleTransformer and decoder): images/s at bs=32, 256x256, fp32 and bf16 (hybrid) modes"""
import os, sys, time, warnings, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mastermetastyletransfer_amd import MasterStyleTransferModel, GraphedStylizer
dev = torch.device("cuda:0")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = MasterStyleTransferModel(swin_variant="swin_T", style_encoder_dim=192, style_decoder_dim=192, style_encoder_num_heads=6, style_decoder_num_heads=6,
                                 style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True).to(dev).eval()
c = torch.rand(32, 3, 256, 256, device=dev); s = torch.rand(32, 3, 256, 256, device=dev)
for dt in (torch.float32, torch.bfloat16):
    m.set_compute_dtype(dt)
    g = GraphedStylizer(m, 32, 256, 256, 1)
    g.content.copy_(c); g.style.copy_(s)
    for _ in range(3): g.graph.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): g.graph.replay()
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 20
    print(f"swin_T, 192-wide ST/decoder, {dt}: {32 / dtm:8.0f} images/s ({dtm * 1e3:.2f} ms per bs=32 step)")
