"""Is the device code of the current sources instruction-for-instruction the code of an earlier commit?

    python tools/same_code_as.py <commit> [source basenames, default: every csrc/*.hip that differs from the commit]

For refactors made without a GPU at hand (macro restructuring, experiment variants behind #ifdef): compiles the named
translation units of <commit> (git archive into a temporary directory) and of the working tree with build.py's flags and
compares the gfx950 disassembly (addresses and encodings stripped). Exit code 1 on any difference.
"""

import glob
import importlib.util
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _flags():
    spec = importlib.util.spec_from_file_location("pg_build", os.path.join(ROOT, "pytorch-generative_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.HIPCC, [f for f in mod.FLAGS if not f.startswith("-W")]


def _disasm(hipcc, flags, src, workdir, tag):
    obj = os.path.join(workdir, tag + ".o")
    subprocess.run([hipcc, *flags, "-c", src, "-o", obj], check=True, capture_output=True)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", obj], check=True, capture_output=True, cwd=workdir)
    co = glob.glob(obj + ".0.hipv4-amdgcn-amd-amdhsa--gfx950")
    if not co:
        return []  # host-only translation unit (comm.hip)
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co[0]], check=True, capture_output=True,
                          text=True).stdout
    return [ln.split("//")[0].rstrip() for ln in text.splitlines()[3:]]


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    commit, names = sys.argv[1], sys.argv[2:]
    hipcc, flags = _flags()
    with tempfile.TemporaryDirectory() as tmp:
        old = os.path.join(tmp, "old")
        os.makedirs(old)
        tar = subprocess.run(["git", "archive", commit, "pytorch-generative_amd/csrc", "include"], cwd=ROOT, check=True,
                             capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", old], input=tar, check=True)
        if not names:
            changed = subprocess.run(["git", "diff", "--name-only", commit, "--", "pytorch-generative_amd/csrc"], cwd=ROOT,
                                     check=True, capture_output=True, text=True).stdout.split()
            hdr = any(c.endswith(".h") for c in changed)
            names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "pytorch-generative_amd", "csrc", "*.hip"))
                           if hdr or any(c.endswith(os.path.basename(p)) for c in changed))
        bad = 0
        for n in names:
            a = _disasm(hipcc, flags, os.path.join(old, "pytorch-generative_amd", "csrc", n + ".hip"), tmp, "old_" + n)
            b = _disasm(hipcc, flags, os.path.join(ROOT, "pytorch-generative_amd", "csrc", n + ".hip"), tmp, "new_" + n)
            same = a == b
            bad += not same
            print(f"{n:20s} {len(b):8d} lines  {'identical' if same else 'DIFFERENT'}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
