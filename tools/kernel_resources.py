"""Register / scratch / LDS table of every kernel in libpg_hip.so, from the code objects' own metadata.

    python tools/kernel_resources.py [--out profiles/rNN_kernel_resources.txt] [--max-spill N]

Reads pytorch-generative_amd/build/*.o (device code extracted with `llvm-objdump --offloading`, metadata with
`llvm-readelf --notes`). `pytorch-generative_amd/build.py` calls check() after linking: a kernel whose
`.vgpr_spill_count` exceeds the limit FAILS the build (spilled registers live in scratch = HBM-backed memory: every
access is a round trip the kernel's roofline never planned for)."""

import argparse
import glob
import os
import subprocess
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.environ.get("PG_OBJ_DIR") or os.path.join(ROOT, "pytorch-generative_amd", "build")
LLVM = "/opt/rocm/lib/llvm/bin"


def _metadata(obj):
    for old in glob.glob(obj + ".0.*"):
        os.remove(old)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", obj], check=True, capture_output=True)
    co = glob.glob(obj + ".0.hipv4-amdgcn-amd-amdhsa--gfx950")
    if not co:
        return []
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co[0]], check=True, capture_output=True,
                           text=True).stdout
    for f in glob.glob(obj + ".0.*"):
        os.remove(f)
    start = notes.find("---")
    end = notes.find("\n...", start)
    if start < 0:
        return []
    meta = yaml.safe_load(notes[start + 3:end if end > 0 else None])
    return meta.get("amdhsa.kernels", []) or []


def table():
    rows = []
    for obj in sorted(glob.glob(os.path.join(OBJ, "*.o"))):
        kernels = _metadata(obj)
        names = [k[".name"] for k in kernels]
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                             text=True).stdout.splitlines() if names else []
        for k, d in zip(kernels, dem):
            short = d.replace("(anonymous namespace)::", "").replace("void ", "")
            short = short[:short.find("(")] if "(" in short else short
            rows.append({"file": os.path.basename(obj)[:-2], "kernel": short, "vgpr": k.get(".vgpr_count", 0),
                         "agpr": k.get(".agpr_count", 0), "sgpr": k.get(".sgpr_count", 0),
                         "vgpr_spill": k.get(".vgpr_spill_count", 0), "sgpr_spill": k.get(".sgpr_spill_count", 0),
                         "scratch_B": k.get(".private_segment_fixed_size", 0),
                         "lds_B": k.get(".group_segment_fixed_size", 0),
                         "max_wg": k.get(".max_flat_workgroup_size", 0)})
    return rows


def render(rows):
    head = f"{'file':18} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>6}  kernel"
    lines = [head]
    for r in sorted(rows, key=lambda r: (-r["vgpr_spill"], -r["scratch_B"], r["file"], r["kernel"])):
        lines.append(f"{r['file']:18} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['vgpr_spill']:6d} "
                     f"{r['sgpr_spill']:6d} {r['scratch_B']:7d} {r['lds_B']:6d}  {r['kernel']}")
    return "\n".join(lines) + "\n"


def check(max_spill=0, out=None, verbose=True):
    """Raises RuntimeError if any kernel spills more than `max_spill` vector registers or uses scratch beyond what
    its spills explain; writes the table to `out` when given."""
    rows = table()
    text = render(rows)
    if out:
        with open(out, "w") as f:
            f.write(f"# {len(rows)} kernels of libpg_hip.so (gfx950); tools/kernel_resources.py\n" + text)
    bad = [r for r in rows if r["vgpr_spill"] > max_spill]
    if verbose:
        spilled = [r for r in rows if r["vgpr_spill"] or r["scratch_B"]]
        print(f"[kernel_resources] {len(rows)} kernels, {len(spilled)} with spills or scratch, "
              f"{len(bad)} above the limit of {max_spill} spilled VGPRs")
    if bad:
        raise RuntimeError("kernels spilling vector registers:\n" + render(bad))
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--max-spill", type=int, default=0)
    a = ap.parse_args()
    try:
        check(a.max_spill, a.out)
    except RuntimeError as e:
        print(e)
        sys.exit(1)
