"""`harry IN OUT [-l L] [-a A] [-q Q] [-c] [-f hry|ply] [--ply-ascii]` -- the reference's command line (main.cc:23-123)
on top of the MI355X path.  Additive flags: --profile compat|chunked, --chunk N, --device D."""
from __future__ import annotations

import sys
import time

from . import codec as hc


def sniff(data: bytes, name: str) -> str:
    """formats/unified_reader.h:33-56"""
    if data[:4] == b"\xfa\xff\xaf\xaf":
        return "hry"
    if data[:3] == b"ply":
        return "ply"
    raise RuntimeError("Not a mesh file")


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    pos, quant_flags, fmt, ascii_, profile, chunk, device = [], [], None, False, "compat", 0, 0
    it = iter(argv)
    for a in it:
        if a in ("-h", "--help"):
            print(__doc__)
            return 0
        if a in ("-f", "--format"):
            fmt = next(it)
        elif a == "--ply-ascii":
            ascii_ = True
        elif a == "--profile":
            profile = next(it)
        elif a == "--chunk":
            chunk = int(next(it))
        elif a == "--device":
            device = int(next(it))
        elif a in ("-l", "--list", "-a", "--attr", "-q", "--quant"):
            quant_flags += [a, next(it)]
        elif a in ("-c", "--clear-quant") or (a[:2] in ("-l", "-a", "-q") and a[2:].isdigit()):
            quant_flags.append(a)
        elif a.startswith("-"):
            print(f"Unknown option {a}", file=sys.stderr)
            return 1
        else:
            pos.append(a)
    if len(pos) != 2:
        print("usage: harry INPUT OUTPUT [options]", file=sys.stderr)
        return 1
    src, dst = pos
    quants, clear = hc.parse_quant_flags(quant_flags)
    out_type = fmt or ("hry" if dst.lower().endswith(".hry") else "ply" if dst.lower().endswith(".ply") else None)
    if out_type not in ("hry", "ply"):
        print("Currently unimplemented", file=sys.stderr)
        return 1
    cx = hc.Codec(device)
    print("Reading input...")
    t0 = time.perf_counter()
    data = open(src, "rb").read()
    kind = sniff(data, src)
    mesh = hc.Mesh.from_ply(data) if kind == "ply" else cx.read_hry(data)
    t1 = time.perf_counter()
    print(f"Reading input took {int((t1 - t0) * 1e3)} ms.")
    if quants or clear:
        print("Quantization...")
        cx.requant(mesh, quants, clear)
        t2 = time.perf_counter()
        print(f"Quantization took {int((t2 - t1) * 1e3)} ms.")
    t2 = time.perf_counter()
    print("Writing output...")
    if out_type == "hry":
        out = cx.write_hry(mesh, profile=hc.PROFILE_CHUNKED if profile == "chunked" else hc.PROFILE_COMPAT, chunk_syms=chunk)
    else:
        out = mesh.to_ply(ascii_)
    open(dst, "wb").write(out)
    t3 = time.perf_counter()
    print(f"Writing output took {int((t3 - t2) * 1e3)} ms.")
    print(f"Total compression time: {int((t3 - t0) * 1e3)} ms")
    print(f"Total input size: {len(data)} Bytes")
    print(f"Total output size: {len(out)} Bytes")
    cx.close()
    return 0


if __name__ == "__main__":
    try:
        sys.exit(main())
    except (hc.HryError, RuntimeError, ValueError) as e:
        print(f"terminate called after throwing an instance of 'std::runtime_error'\n  what():  {getattr(e, 'msg', e)}", file=sys.stderr)
        sys.exit(134)
