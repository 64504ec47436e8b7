"""Kernel-name normalisation for the profile scripts.  rocprofv3 (ROCm 7.2) leaves names whose template arguments
include __bf16 mangled (`_ZN4lshm12recon_kernelILb1ELb1EDF16bEEvPKf...`): rebuild the readable form
`recon_kernel<true, true, __bf16>` from the Itanium mangling of the simple cases used here (bool / int literals,
float, __bf16)."""
import re


def short(name: str) -> str:
    n = name.strip('"')
    if n.startswith("_ZN4lshm"):
        m = re.match(r"_ZN4lshm(\d+)", n)
        ln = int(m.group(1))
        ident = n[m.end():m.end() + ln]
        rest = n[m.end() + ln:]
        args = []
        if rest.startswith("I"):
            i = 1
            while i < len(rest) and rest[i] != "E":
                if rest.startswith("Lb", i):
                    args.append("true" if rest[i + 2] == "1" else "false"); i += 4
                elif rest.startswith("Li", i):
                    j = rest.index("E", i); args.append(rest[i + 2:j].replace("n", "-")); i = j + 1
                elif rest.startswith("DF16b", i):
                    args.append("__bf16"); i += 5
                elif rest[i] == "f":
                    args.append("float"); i += 1
                else:
                    return ident + "<?>"
        return ident + ("<" + ", ".join(args) + ">" if args else "")
    # (rocprofv3's own demangler renders the vendor type DF16b of some instantiations as "bool _Accum")
    return n.replace("(anonymous namespace)::", "").replace("lshm::", "").replace("void ", "").replace("bool _Accum", "__bf16").split("(")[0]
