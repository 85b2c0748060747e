"""ONE process, PixelSNAIL (bench constructor, batch 32, bit-reproducible kernels), 6 eager steps: with a device synchronisation after every
phase against fully asynchronous steps — the parameters must be bit-identical (profiles/README.md round 5 item 16)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402

me = os.path.join(ROOT, "tools", "exp", "two_proc_dp_divergence.py")
tmp = "/tmp/svn"
os.makedirs(tmp, exist_ok=True)
model = sys.argv[1] if len(sys.argv) > 1 else "pixel_snail"
for tag, env in (("sync", {}), ("nosync", {"NOSYNC": "1", "FAKE_STAGING": os.environ.get("FAKE_STAGING", "0")}),
                 ("nosync2", {"NOSYNC": "1", "FAKE_STAGING": os.environ.get("FAKE_STAGING", "0")})):
    subprocess.run([sys.executable, me, "worker", f"{tmp}/{tag}.pt", "1", "0", "0"], check=True, env=dict(os.environ, **env))
ref = torch.load(f"{tmp}/sync.pt")
for tag in ("nosync", "nosync2"):
    d = torch.load(f"{tmp}/{tag}.pt")
    for k in d["order"]:
        if k not in ref["rec"]:
            continue
        a, b = d["rec"][k], ref["rec"][k]
        n = int((a != b).sum())
        print(f"{tag} {k}: {n} of {a.numel()} elements differ from the synchronised run" + (f", max |diff| {float((a - b).abs().max()):.3e}" if n else ""))
