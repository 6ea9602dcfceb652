#!/usr/bin/env python3
"""Instruction mix of the kernels of one .hip file (static count over the whole body, loops counted once): compile with --save-temps, read the .s.
    python scripts/isa_count.py lc_amd/csrc/lc_dense_aux.hip [name filter]"""
import os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-c", src, "-I" + os.path.join(root, "include"), "--save-temps"] + sys.argv[3:],
               cwd=d, check=True, capture_output=True)
s = open([os.path.join(d, f) for f in os.listdir(d) if f.endswith("gfx950.s")][0]).read()
for m in re.finditer(r"^(_Z\S+): +; @\S+\n(.*?)\n\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name or ".amdhsa_kernel " + name not in s:
        continue
    ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    vg = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", s)
    cnt = lambda *pre: sum(1 for i in ins if i.startswith(pre))
    print(f"{name[:100]}\n   total {len(ins)} valu {cnt('v_')} global {cnt('global_', 'buffer_')} lds {cnt('ds_')} trans {cnt('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')} "
          f"div {cnt('v_div')} cvt {cnt('v_cvt')} pk {cnt('v_pk')} salu {cnt('s_')} vgpr {vg.group(1) if vg else '?'}")
