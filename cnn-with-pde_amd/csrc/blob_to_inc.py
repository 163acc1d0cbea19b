#!/usr/bin/env python3
"""Write code objects as C byte arrays plus a table: blob_to_inc.py out.inc spec=file [spec=file ...]
(spec = "<waves>" or "<waves>a<bits>", see gen_adi_bwd_asm.py)."""
import sys

if "--fwd" in sys.argv:                      # the forward kernels' table: out.inc --fwd <waves>[t]=file ...
    sys.argv.remove("--fwd")
    with open(sys.argv[1], "w") as out:
        rows = []
        for spec in sys.argv[2:]:
            name, path = spec.split("=", 1)
            data = open(path, "rb").read()
            out.write(f"alignas(4096) static const unsigned char kAsmFwdBlob_{name}[{len(data)}] = {{\n")
            for i in range(0, len(data), 32):
                out.write(",".join(str(b) for b in data[i:i + 32]) + ",\n")
            out.write("};\n")
            rows.append(f'    {{"{name}", {int(name.rstrip("t"))}, kAsmFwdBlob_{name}, "adi_fwd_asm_n32_w{name}", {{}}, {{}}}},')
        out.write("static Variant g_fwd_variants[] = {\n" + "\n".join(rows) + "\n};\n")
    sys.exit(0)

with open(sys.argv[1], "w") as out:
    rows = []
    for spec in sys.argv[2:]:
        name, path = spec.split("=", 1)
        data = open(path, "rb").read()
        nw = int(name.split("a")[0].rstrip("b"))
        out.write(f"alignas(4096) static const unsigned char kAsmBlob_{name}[{len(data)}] = {{\n")
        for i in range(0, len(data), 32):
            out.write(",".join(str(b) for b in data[i:i + 32]) + ",\n")
        out.write("};\n")
        rows.append(f'    {{"{name}", {nw}, kAsmBlob_{name}, "adi_bwd_asm_n32_w{name}", {{}}, {{}}}},')
    out.write("static Variant g_variants[] = {\n" + "\n".join(rows) + "\n};\n")
