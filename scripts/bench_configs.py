"""Secondary measurements: one-GPU step time of the other BASELINE configs (parity-test cases, not bench lines).
    python scripts/bench_configs.py [3 4 5]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dinov2_distillation_amd.trainer import DistillationStepper

CONFIGS = {
    "1": dict(teacher="dinov2_vits14", student="stdc_2", crop=224, batch=2, heads=(16, 24)),
    "2": dict(teacher="dinov2_vits14", student="convnext_tiny", crop=224, batch=64, heads=(16, 24)),
    "3": dict(teacher="dinov2_vitb14", student="resnet_50", crop=224, batch=128, heads=(16, 24)),
    "4": dict(teacher="dinov2_vitl14", student="swin_tiny", crop=518, batch=16, heads=(16, 32)),
    "5": dict(teacher="dinov2_vitg14", student="convnext_base", crop=224, batch=32, heads=(16, 24), teacher_precision="bf16"),
    "presnet50": dict(teacher="dinov2_vits14", student="presnet_50", crop=224, batch=64, heads=(16, 24)),
    "mobilenet_v2": dict(teacher="dinov2_vits14", student="mobilenet_v2", crop=224, batch=64, heads=(16, 24)),
    "darknet_m": dict(teacher="dinov2_vits14", student="darknet_m", crop=224, batch=64, heads=(16, 24)),
    "mit_b2": dict(teacher="dinov2_vits14", student="mit_b2", crop=224, batch=64, heads=(16, 24)),
    "2a": dict(teacher="dinov2_vits14", student="convnext_tiny", crop=224, batch=64, heads=(16, 24), teacher_precision="bf16_autocast"),
    "3a": dict(teacher="dinov2_vitb14", student="resnet_50", crop=224, batch=128, heads=(16, 24), teacher_precision="bf16_autocast"),
    "4a": dict(teacher="dinov2_vitl14", student="swin_tiny", crop=518, batch=16, heads=(16, 32), teacher_precision="bf16_autocast"),
    "5a": dict(teacher="dinov2_vitg14", student="convnext_base", crop=224, batch=32, heads=(16, 24), teacher_precision="bf16_autocast"),
    "2as": dict(teacher="dinov2_vits14", student="convnext_tiny", crop=224, batch=64, heads=(16, 24), teacher_precision="bf16_autocast",
                student_precision="bf16_autocast"),
    "5as": dict(teacher="dinov2_vitg14", student="convnext_base", crop=224, batch=32, heads=(16, 24), teacher_precision="bf16_autocast",
                student_precision="bf16_autocast"),
    "5p": dict(teacher="dinov2_vitg14", student="convnext_base", crop=224, batch=32, heads=(16, 24), precision="bf16_autocast"),
    "2p": dict(teacher="dinov2_vits14", student="convnext_tiny", crop=224, batch=64, heads=(16, 24), precision="bf16_autocast"),
    "mobilenet_v3_large": dict(teacher="dinov2_vits14", student="mobilenet_v3_large", crop=224, batch=64, heads=(16, 24)),
    "5f32": dict(teacher="dinov2_vitg14", student="convnext_base", crop=224, batch=32, heads=(16, 24)),
}

for name in sys.argv[1:] or ["3", "4", "5"]:
    c = CONFIGS[name]
    st = DistillationStepper(teacher=c["teacher"], student=c["student"], crop=c["crop"], heads=c["heads"], device="cuda:0",
                             teacher_precision=c.get("teacher_precision"), student_precision=c.get("student_precision"),
                             precision=c.get("precision"))
    x = torch.randn(c["batch"], 3, c["crop"], c["crop"], generator=torch.Generator().manual_seed(1234)).cuda()
    for _ in range(2):
        loss = st.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        loss = st.step(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"config": name, **{k: v for k, v in c.items()}, "ms_per_step": round(dt * 1e3, 2),
                      "images_per_s": round(c["batch"] / dt, 2), "loss": round(float(loss), 5),
                      "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}), flush=True)
    del st, x
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
