#!/usr/bin/env python3
"""Write binary files as C byte arrays: blob_to_inc.py out.inc name=file [name=file ...] (used for the embedded code objects)."""
import sys

with open(sys.argv[1], "w") as out:
    for spec in sys.argv[2:]:
        name, path = spec.split("=", 1)
        data = open(path, "rb").read()
        out.write(f"alignas(4096) static const unsigned char {name}[{len(data)}] = {{\n")
        for i in range(0, len(data), 32):
            out.write(",".join(str(b) for b in data[i:i + 32]) + ",\n")
        out.write("};\n")
