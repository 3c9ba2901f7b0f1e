"""Static resource table of every kernel in repsurf_amd/lib/librepsurf_hip.so: VGPRs / AGPRs / SGPRs, LDS bytes, scratch bytes,
spill counts, occupancy bound by registers -- read from the code objects' metadata notes (no GPU needed).

    python tools/kernel_resources.py [lib.so] > profiles/r05/kernel_resources.csv

The .hip_fatbin section holds one clang offload bundle per translation unit; every bundle's gfx950 entry is an ELF whose
NT_AMDGPU_METADATA note (msgpack) lists the kernels.  llvm-readelf prints it as YAML."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    pos = 0
    while True:
        at = blob.find(MAGIC, pos)
        if at < 0:
            return
        p = at + len(MAGIC)
        (n,) = struct.unpack_from("<Q", blob, p)
        p += 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                yield blob[at + off:at + off + size]
        pos = at + len(MAGIC)


def kernels_of(elf_bytes):
    import yaml
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        text = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    out = []
    for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.", text, re.S | re.M):
        meta = yaml.safe_load(doc) or {}
        for k in meta.get("amdhsa.kernels", []):
            out.append({key.lstrip("."): val for key, val in k.items() if key != ".args"})
    return out


def demangle(names):
    r = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(anonymous namespace\)::", "", x) for x in r]


UNITS = ("fps", "ballquery", "knn_umbrella", "knn_wide", "group", "interp", "seg_geom", "scene_knn", "grid_knn", "mlp (fp32 MFMA)", "umbrella_mlp",
         "umbrella_mfma", "head", "adam", "mlp_bf16 (bf16 operands)", "mlp_sb (bf16 storage)", "mlp_split (3 x bf16 split products)")      # link order (Makefile OBJS)


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "repsurf_amd", "lib", "librepsurf_hip.so")
    blob = open(lib, "rb").read()
    rows = []
    for u, co in enumerate(code_objects(blob)):
        for k in kernels_of(co):
            k["unit"] = UNITS[u] if u < len(UNITS) else str(u)
            rows.append(k)
    names = demangle([k["name"] for k in rows])
    print("unit,kernel,vgprs,agprs,sgprs,lds_bytes,scratch_bytes,vgpr_spills,sgpr_spills,max_workgroup,waves_per_simd_by_registers")
    for k, nm in sorted(zip(rows, names), key=lambda x: (x[0]["unit"], x[1])):
        nm = re.sub(r"\(.*$", "", nm.replace("void ", ""))
        v, a = int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0))
        total = -(-v // 8) * 8 + a if a else v         # unified register file: 512 per SIMD lane, allocated in blocks of 8
        waves = min(8, 512 // max(8, -(-total // 8) * 8))
        print(",".join(str(x) for x in (k["unit"], '"%s"' % nm, v, a, k.get("sgpr_count", ""), k.get("group_segment_fixed_size", ""),
                                        k.get("private_segment_fixed_size", ""), k.get("vgpr_spill_count", ""), k.get("sgpr_spill_count", ""),
                                        k.get("max_flat_workgroup_size", ""), waves)))


if __name__ == "__main__":
    main()
