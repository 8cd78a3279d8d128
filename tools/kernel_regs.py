"""Register / scratch usage per kernel from the gfx950 assembly of one .hip file: python tools/kernel_regs.py comfy-rvc_amd/csrc/conv_x3.hip [filter]"""
import re, subprocess, sys, tempfile, os
src = os.path.abspath(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{root}/include", f"-I{os.path.dirname(src)}",
                    "-save-temps", "-c", src, "-o", "x.o"] + sys.argv[3:], cwd=d, stderr=subprocess.DEVNULL, check=True)
    asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    s = open(os.path.join(d, asm)).read()
for blk in s.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if name.startswith("_Z"):
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name: continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print(f"vgpr {g('vgpr_count'):>4} agpr {blk.split(chr(10))[0].strip():>4} sgpr {g('sgpr_count'):>4} spill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>5}  {name[:100]}")
