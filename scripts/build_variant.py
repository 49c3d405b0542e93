#!/usr/bin/env python
"""A/B builds: `python scripts/build_variant.py NAME file.hip[,file2.hip] -DFOO=1 ...` compiles the named translation units with the extra flags, links them with
the other units' objects of the regular build (m-loam_amd/build/) into m-loam_amd/lib_ab/NAME/libmloam_hip.so. Select one with MLOAM_HIP_LIB=<path> (a debug /
measurement switch of the python harness; the product loads m-loam_amd/lib/libmloam_hip.so). Variants travel to the GPU box like every built .so."""
import importlib.util, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("mloam_build", os.path.join(ROOT, "m-loam_amd", "build.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
name, files, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build()
out = os.path.join(ROOT, "m-loam_amd", "lib_ab", name)
os.makedirs(out, exist_ok=True)
objs = []
for s in B.SOURCES:
    obj = os.path.join(B.OBJDIR, s.replace(".hip", ".o"))
    if s in files:
        obj = os.path.join(out, s.replace(".hip", ".o"))
        subprocess.run([B._hipcc()] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, s), "-o", obj], check=True)
    objs.append(obj)
lib = os.path.join(out, "libmloam_hip.so")
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"], check=True)
print(lib)
