"""Per-class instruction histogram of one loop of one kernel, from hipcc's gfx950 assembly.

  python tools/isa_histogram.py [--asm FILE.s] [--kernel anim_postnuc_fwd_kernel] [--tag pn_diag_advanceILi4ELi1ELb0ELb0] [--out profiles/...txt]

Without --asm the tool compiles pyani_amd/csrc/pg_anim.hip with `--save-temps` into a temporary directory (about a minute).  The loop
is found by --tag: the innermost loop whose blocks carry that inlined-function tag in the compiler's block comments (default: the
un-forced DPL = 4 step of the diagonal engine, pga_postnuc_diag.inc pn_diag_advance<4, 1, false, false>).  Output: every basic block
of that loop with its instruction classes, blocks of the rare paths (window refill, window move, wave-wide maximum) marked by the
inlined function their label names, and the totals of the straight path (the blocks without such a mark).
"""
import argparse
import re
import subprocess
import sys
import tempfile
from collections import Counter
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def classify(op):
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "wait/nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_barrier")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("ds_",)):
        return "LDS"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "VMEM"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "VALU lane<->scalar"
    if op.startswith("v_cmp"):
        return "VALU compare"
    if op.startswith("v_cndmask"):
        return "VALU select"
    if op.startswith("v_"):
        return "VALU"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--kernel", default="anim_postnuc_fwd_kernel")
    ap.add_argument("--tag", default="pn_diag_advanceILi4ELi1ELb0ELb0")
    ap.add_argument("--out")
    a = ap.parse_args()
    if a.asm:
        text = Path(a.asm).read_text()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            csrc = ROOT / "pyani_amd" / "csrc"
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", f"-I{ROOT / 'include'}",
                            f"-I{csrc}", "-c", str(csrc / "pg_anim.hip"), "--save-temps", "-o", f"{tmp}/pg_anim.o"], check=True, cwd=tmp)
            text = next(Path(tmp).glob("*gfx950.s")).read_text()
    lines = text.splitlines()
    # the kernel's body
    start = next(i for i, ln in enumerate(lines) if re.match(rf"^_ZN.*{a.kernel}.*:\s", ln))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or re.match(r"^_ZN.*:\s", lines[i]) or lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    # blocks: (label, comment, header, instructions)
    blocks, cur = [], {"label": "entry", "comment": "", "header": None, "ins": []}
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln) or re.match(r"^; %bb\.(\d+):\s*(;.*)?$", ln)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "comment": (m.group(2) or ""), "header": None, "ins": []}
            h = re.search(r"Header=(BB\d+_\d+)", ln)
            cur["header"] = h.group(1) if h else None
            continue
        if ln.startswith("\t") and not ln.lstrip().startswith((";", ".")):
            cur["ins"].append(ln.split()[0])
            cur.setdefault("text", []).append(ln.split(";")[0].rstrip())
        elif "This Inner Loop Header" in ln or "This Loop Header" in ln:
            cur["is_header"] = True
        h = re.search(r"in Loop: Header=(BB\d+_\d+)", ln)
        if h and cur["header"] is None:
            cur["header"] = h.group(1)
    blocks.append(cur)
    tagged = [b for b in blocks if a.tag in b["comment"]]
    if not tagged:
        sys.exit(f"no block of {a.kernel} carries the tag {a.tag}")
    header = Counter(b["header"] for b in tagged if b["header"]).most_common(1)[0][0]
    loop = [b for b in blocks if b["header"] == header or b["label"].lstrip(".L") == header]
    rare_marks = ("pn_diag_refill", "packed_window", "window_check", "window_move", "wave_max", "update_best", "rev_window", "strand_window", "Flow")
    out = [f"kernel {a.kernel}, loop with header {header} (blocks tagged {a.tag}): {len(loop)} basic blocks, {sum(len(b['ins']) for b in loop)} instructions",
           "classes: VALU (arithmetic / logic / DPP moves), VALU compare, VALU select, VALU lane<->scalar (readlane / readfirstlane), SALU, SMEM, VMEM, LDS, branch, wait/nop", ""]
    straight, rare, cells, cell_blocks = Counter(), Counter(), Counter(), []
    for b in loop:
        c = Counter(classify(op) for op in b["ins"])
        if not c:
            continue
        fn = re.search(r"_ZN[\w]*?(pn_diag_\w+?|packed_window_\w+?|diag_\w+?)I?L?i?", b["comment"])
        mark = next((m for m in rare_marks if m in b["comment"]), None)
        dpp = sum(1 for op in b["ins"] if op.endswith("_dpp"))
        if any("wave_sh" in t for t in b.get("text", [])):
            mark = None
            cells.update(c)
            cell_blocks.append(b)
        (rare if mark else straight).update(c)
        out.append(f"{b['label']:>12} {'[' + mark + ']' if mark else '':<18} {sum(c.values()):4d}  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items())) + (f"  (DPP {dpp})" if dpp else ""))
    out += ["", "straight path (blocks without a rare-path mark): " + "  ".join(f"{k} {v}" for k, v in sorted(straight.items())) + f"  = {sum(straight.values())}",
            "rare paths (refill / window move / wave-wide maximum / compiler flow blocks): " + "  ".join(f"{k} {v}" for k, v in sorted(rare.items())) + f"  = {sum(rare.values())}",
            "(one trip of this loop is TWO anti-diagonals: the odd and the even parity bodies)", "",
            "the two CELL blocks (the ones holding the wave-shift DPP pair: two cells per lane each, pgd::diag_lane_step): " + "  ".join(f"{k} {v}" for k, v in sorted(cells.items())) + f"  = {sum(cells.values())}"]
    for b in cell_blocks:
        out += ["", f"---- {b['label']} ----"] + b.get("text", [])
    txt = "\n".join(out) + "\n"
    if a.out:
        Path(a.out).write_text(txt)
    print(txt)


if __name__ == "__main__":
    main()
