"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol include/ovla.h
declares, and the Python structs agree with the header (no compute calls: there is no GPU here)."""
import ctypes
import importlib
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def libmod(pkg):
    import __graft_entry__ as g

    g.build()
    return importlib.import_module("openvla-oft_amd._lib")


def test_every_declared_symbol_is_exported(libmod):
    handle = libmod.lib()
    names = sorted(libmod.FUNCTIONS)
    assert len(names) >= 30
    for n in names:
        assert hasattr(handle, n), f"libovla_hip.so does not export {n}"
    out = subprocess.run(["nm", "-D", str(libmod.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(names) <= exported
    extra = {e for e in exported if e.startswith("ovla_")} - set(names)
    assert not extra, f"exported but undeclared: {extra}"


def test_abi_version_and_error_channel(libmod):
    handle = libmod.lib()
    assert handle.ovla_abi_version() == 1
    args = libmod.STRUCTS["ovla_gemm_args"]()  # all zero: must be rejected before any launch
    rc = handle.ovla_gemm_bf16(ctypes.byref(args), None)
    assert rc == -1
    assert b"ovla_gemm_bf16" in handle.ovla_last_error()


def test_struct_layout_matches_header(libmod):
    # spot-check offsets the C compiler would produce (natural alignment) for the most used struct
    S = libmod.STRUCTS["ovla_gemm_args"]
    assert S.A.offset == 0 and S.lda.offset == 8 and S.B.offset == 16
    assert S.M.offset % 4 == 0 and ctypes.sizeof(S) % 8 == 0
    # compile a tiny C program against the header and compare sizeof for every struct
    src = ["#include <stdio.h>", f'#include "{ROOT / "include" / "ovla.h"}"', "int main(){"]
    for name in libmod.STRUCTS:
        src.append(f'printf("{name} %zu\\n", sizeof({name}));')
    src.append("return 0;}")
    exe = ROOT / "openvla-oft_amd" / "_build" / "abi_sizes"
    exe.parent.mkdir(exist_ok=True)
    c = exe.with_suffix(".c")
    c.write_text("\n".join(src))
    subprocess.run(["gcc", "-o", str(exe), str(c)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        name, size = line.split()
        assert ctypes.sizeof(libmod.STRUCTS[name]) == int(size), name


def test_missing_library_fails_loudly(libmod, monkeypatch, tmp_path):
    monkeypatch.setattr(libmod, "_lib", None)
    monkeypatch.setattr(libmod, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        libmod.lib()


def test_integration_md_struct_matches_header():
    """The ctypes struct INTEGRATION.md shows a maintainer must have the header's field order (a stale snippet would corrupt arguments)."""
    import re
    from pathlib import Path

    lib_mod = importlib.import_module("openvla-oft_amd._lib")
    text = (Path(__file__).resolve().parent.parent / "INTEGRATION.md").read_text()
    block = text[text.index("class ovla_gemm_args(ctypes.Structure):"): text.index("def linear_bf16(")]
    doc_fields = re.findall(r'\("(\w+)", ctypes\.(\w+)\)', block)
    kinds = {"c_void_p": "c_void_p", "c_int64": "c_long", "c_int32": "c_int", "c_float": "c_float"}
    hdr = [(n, t.__name__) for n, t in lib_mod.STRUCTS["ovla_gemm_args"]._fields_]
    assert [n for n, _ in doc_fields] == [n for n, _ in hdr]
    assert all(kinds[d] == h or (d == "c_int64" and h in ("c_long", "c_longlong")) for (_, d), (_, h) in zip(doc_fields, hdr))


def test_shipped_code_object_has_no_packed_fp32(libmod, tmp_path):
    """The product library must be free of the compiler's packed-FP32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 /
    v_pk_mov_b32): a gfx950 hazard drops a term of such an accumulate when another stream's waves share the SIMD (DESIGN.md "Run-to-run
    determinism", profiles/r02_norm_bwd_probe.md).  csrc/build.sh switches the target feature off for EVERY .hip file it compiles; this test
    disassembles every gfx950 code object inside the built .so so that a new source file, a changed flag or another build mode cannot lose
    the protection silently."""
    import re
    import shutil

    objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        pytest.skip("llvm-objdump not in this image")
    so = tmp_path / "libovla_hip.so"
    shutil.copy(libmod.LIB_PATH, so)
    subprocess.run([str(objdump), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)   # extracts the bundles next to `so`
    objs = sorted(tmp_path.glob("libovla_hip.so.*gfx950*"))
    n_src = len(list((ROOT / "openvla-oft_amd" / "csrc").glob("*.hip")))
    assert 1 <= len(objs) <= n_src, f"{len(objs)} gfx950 code objects for {n_src} .hip sources"
    bad, n_mfma = [], 0
    pat = re.compile(r"\bv_pk_(add_f32|mul_f32|fma_f32|mov_b32)\b")
    for o in objs:
        text = subprocess.run([str(objdump), "-d", str(o)], check=True, capture_output=True, text=True).stdout
        n_mfma += text.count("v_mfma_")
        bad += [f"{o.name}: {ln.strip()}" for ln in text.splitlines() if pat.search(ln)][:5]
    assert n_mfma > 500, "the disassembly must be the real kernels (MFMA instructions present)"
    assert not bad, "packed-FP32 instructions in the shipped code object:\n" + "\n".join(bad)
    # and build.sh compiles every .hip file of the directory with the flag
    sh = (ROOT / "openvla-oft_amd" / "csrc" / "build.sh").read_text()
    assert 'SRCS="$(ls *.hip' in sh and 'NOPK="$SRCS"' in sh


def test_w4_gemm_loop_is_what_the_source_wrote(libmod, tmp_path):
    """The 4-wave 256x256 GEMM's K loop is inline asm with hand-counted `s_waitcnt`s (gemm_nt.hip: gemm_nt_w4_kernel).  That is only sound while the
    compiler adds nothing of its own between the asm statements: a register copy of an asm load's destination would read it before the wait, a spill would
    change the vmcnt arithmetic, a v_readfirstlane feeding an asm load's SGPR base needs wait states the compiler cannot insert.  This test disassembles
    the shipped kernels and checks the loop bodies: per K tile 128 MFMAs, 32 ds_read_b128, 16 ds_write_b128, 16 global_load_dwordx4 -- and NO other vector
    ALU instruction, no AGPR move, no scratch access, no readfirstlane."""
    import re
    import shutil

    objdump = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")
    if not objdump.exists():
        pytest.skip("llvm-objdump not in this image")
    so = tmp_path / "libovla_hip.so"
    shutil.copy(libmod.LIB_PATH, so)
    subprocess.run([str(objdump), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    seen = 0
    for o in sorted(tmp_path.glob("libovla_hip.so.*gfx950*")):
        text = subprocess.run([str(objdump), "-d", str(o)], check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r"<(_ZN\S*gemm_nt_w4_kernel\S*)>:\n(.*?)(?=\n\n|\Z)", text, re.S):
            ins = [ln.split("//")[0].split("\t")[1].strip() if "\t" in ln else ln.strip() for ln in m.group(2).splitlines() if ln.strip()]
            ins = [i for i in ins if i and not i.endswith(":")]
            ops = [i.split()[0] for i in ins]
            barriers = [k for k, op in enumerate(ops) if op == "s_barrier"]
            # the K loop = the stretch between the first and the last s_barrier that is followed by MFMAs (main loop x2 bodies + the odd tail body);
            # the epilogue's barrier (__syncthreads) comes after the last MFMA
            last_mfma = max(k for k, op in enumerate(ops) if op.startswith("v_mfma"))
            body_starts = [k for k in barriers if k < last_mfma]
            seen += 1
            assert len(body_starts) == 3, f"{m.group(1)}: {len(body_starts)} K-loop bodies (two unrolled + the odd tail expected)"
            # a body = from its barrier to its last MFMA (what lies between two bodies -- loop control, the branch to the tail -- is the compiler's)
            ends = [max(k for k in range(b, nxt) if ops[k].startswith("v_mfma")) for b, nxt in zip(body_starts, body_starts[1:] + [last_mfma + 1])]
            loop = [op for b, e in zip(body_starts, ends) for op in ops[b:e + 1]]
            c = lambda pre: sum(1 for op in loop if op.startswith(pre))
            wide = "Li2ELb" in m.group(1)        # <KEXT, ABL, WNW = 2, RMAP, GMAP>: the 256x256 tile (128x128 per wave); WNW = 4: 128x256 (128x64 per wave)
            nt, npieces = (8, 16) if wide else (4, 12)
            assert c("v_mfma") == 3 * 16 * nt + nt, (m.group(1), c("v_mfma"))                      # + the last deferred row
            assert c("ds_read_b128") == 3 * (16 + 2 * nt) and c("ds_write_b128") == 2 * npieces and c("global_load_dwordx4") == 2 * npieces, (m.group(1), c("ds_read_b128"), c("ds_write_b128"), c("global_load_dwordx4"))
            alien = [op for op in loop if op.startswith(("v_", "scratch_", "buffer_", "flat_")) and not op.startswith("v_mfma")]
            assert not alien, f"{m.group(1)}: the compiler put {sorted(set(alien))} inside the hand-scheduled K loop"
            # everywhere in the kernel (the K-extension prologue included): no compiler-generated vector ALU instruction may WRITE a register an asm MFMA
            # reads within the three instructions before it -- the hazard recognizer does not see that the asm reads the register (a v_mov right before the
            # first K-extension MFMA once fed it a stale fragment); the source puts `s_nop 4` between compiler code and every MFMA sequence
            def regs(tok):
                mm = re.match(r"([va])\[(\d+):(\d+)\]", tok) or re.match(r"([va])(\d+)$", tok)
                if not mm:
                    return set()
                lo = int(mm.group(2)); hi = int(mm.group(3)) if mm.lastindex == 3 else lo
                return {(mm.group(1), r) for r in range(lo, hi + 1)}
            for k, line in enumerate(ins):
                if line.startswith("v_mfma"):
                    toks = [x.strip() for x in line.split(None, 1)[1].split(",")]
                    src = regs(toks[1]) | regs(toks[2])
                    for back in range(1, 4):
                        if k - back < 0:
                            break
                        prev = ins[k - back]
                        if prev.startswith("s_nop"):
                            break
                        if prev.startswith("v_") and not prev.startswith("v_mfma"):
                            dst = regs(prev.split(None, 1)[1].split(",")[0].strip())
                            assert not (dst & src), f"{m.group(1)}: `{prev}` writes a source of `{line}` {back} instruction(s) before it"
    assert seen == 10, f"{seen} gemm_nt_w4_kernel instantiations found in the shipped library (256x256: K-extensions of 0 / 32 / 64 / 96 columns, 0 / 32 with the SwiGLU pair map; 128x256: 0, 32, 0 with the RoPE column map, 0 with the SwiGLU pair map)"
