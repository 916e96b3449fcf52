#!/usr/bin/env python3
"""Checks the `@reads` / `@writes` / `@exports` lists at the top of each csrc/bsx_step_phase_*.inl against the file's own text.

The seven phase files are textual units of ONE kernel body (bsx_step_kernel.h includes them in tick order inside the tick loop); they share
the kernel's locals instead of passing ~40 values by reference.  What keeps them reviewable in isolation is a contract per file, and
this tool is what makes the contract more than a comment (it runs in the CPU suite: tests/test_host_cpu.py):

    // @reads    names that EARLIER phases export and this phase uses
    // @writes   names declared OUTSIDE this file (kernel locals of bsx_step_kernel.h, or an earlier phase's exports) that this phase assigns
    // @exports  names this phase declares at phase scope (brace depth 0 of the file) and a LATER phase, or the kernel after the includes, uses
    // @lds      LDS arrays (s_*) this phase stores to (plain stores and atomics)

Checked, per file:  every outside name the text assigns is in @writes (nothing is modified behind the reader's back); every phase-scope
declaration that a later file uses is in @exports; every @reads name is exported by an earlier phase and used here, and every earlier
export used here is in @reads; every s_* array stored to is in @lds; no list carries a stale name.
What it cannot see: writes through a reference parameter of a helper (the helpers that take one are listed in BY_REF below) and stores
to global memory (those are the `stores` phase's business and named in its prose).  It is a text scanner, not a compiler: it strips
comments and strings, tracks brace depth, and recognises `Type name`, `Type name = ...`, and assignment operators.

    python tools/check_phase_contract.py            # check, exit 1 on any violation
    python tools/check_phase_contract.py --print    # print what the scanner derives per file (to write or update the lists)
    python tools/check_phase_contract.py --csrc DIR # the same on another copy of the sources"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deep-rl-battlespace_amd", "csrc")
PHASES = ["actor", "shot", "move", "geometry", "bullets", "outcome", "stores"]
NOT_TYPES = {"return", "else", "case", "typedef", "struct", "goto", "new", "delete", "typename", "sizeof", "using", "namespace", "if", "for",
             "while", "do", "switch", "template", "class", "enum", "break", "continue", "default", "operator", "defined"}
QUALS = {"const", "constexpr", "static", "volatile", "unsigned", "signed", "long", "short", "inline", "register", "mutable", "__shared__", "__restrict__"}
# helpers that write through reference parameters: name -> 0-based positions of the written arguments
BY_REF = {"unpack_plane": (1, 2, 3, 4), "sincos": (1, 2), "load_inputs": (1,), "atan2_pixels_n": (2,), "geometry_n": ()}
ASSIGN_OPS = ("<<=", ">>=", "+=", "-=", "*=", "/=", "|=", "&=", "^=", "%=", "=")


def strip(text):
    """Comments and string / char literals out (newlines kept)."""
    text = re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)
    return re.sub(r"'(?:\\.|[^'\\])'", "0", text)


def header_lists(raw):
    out = {}
    for tag in ("reads", "writes", "exports", "lds"):
        names = []
        for m in re.finditer(r"^//\s*@%s\b(.*)$" % tag, raw, flags=re.M):
            names += [t for t in re.split(r"[\s,]+", m.group(1).strip()) if t and t != "-"]
        out[tag] = names
    return out


def scan(text):
    """-> dict(decl0: names declared at brace depth 0, local: names declared deeper or inside parentheses, assigned: root names of
    every assignment / increment / by-reference write with the depth it happens at, used: every identifier, lds: s_* arrays stored to)."""
    text = strip(text)
    decl0, local, assigned, lds = set(), set(), set(), set()
    used = set(re.findall(r"\b[A-Za-z_]\w*\b", text))
    # statements: split on ; { } keeping track of depth; parentheses depth decides whether a declaration is a parameter / loop variable
    depth = 0
    paren = 0
    buf = []
    stmts = []                                               # (depth at statement start, text)
    start_depth = 0
    for ch in text:
        if ch == "(":
            paren += 1
        elif ch == ")":
            paren = max(0, paren - 1)
        if ch in ";{}" and (paren == 0 or ch in "{}"):
            stmts.append((start_depth, "".join(buf)))
            buf = []
            if ch == "{":
                depth += 1
            elif ch == "}":
                depth = max(0, depth - 1)
            start_depth = depth
            if ch in "{}":
                paren = 0 if ch == "}" and depth == 0 else paren
            continue
        buf.append(ch)
    stmts.append((start_depth, "".join(buf)))
    ident = r"[A-Za-z_]\w*"
    tmpl = r"(?:<[^;{}]*?>)?"
    decl_re = re.compile(r"(?<![\w.>])((?:(?:%s)\s+)*)(%s(?:::%s)*)%s\s*[\*&]*\s+[\*&]*\s*(%s)\s*(?=[=;,)\[{(]|$)" % ("|".join(sorted(QUALS)), ident, ident, tmpl, ident))

    def split_top(s, sep=","):
        parts, d, cur = [], 0, []
        for ch in s:
            if ch in "([{<" and not (ch == "<" and False):
                d += ch in "([{"
            elif ch in ")]}":
                d -= 1
            if ch == sep and d == 0:
                parts.append("".join(cur)); cur = []
            else:
                cur.append(ch)
        parts.append("".join(cur))
        return parts

    for d, s in stmts:
        s1 = s.strip()
        if not s1:
            continue
        # --- declarations
        for m in decl_re.finditer(s):
            ty, name = m.group(2), m.group(3)
            if ty in NOT_TYPES or name in NOT_TYPES or ty in QUALS and False:
                continue
            # inside parentheses (a parameter, a loop variable, a lambda's argument)?  count unmatched '(' before the match
            before = s[:m.start()]
            in_paren = before.count("(") - before.count(")") > 0
            (local if (in_paren or d > 0) else decl0).add(name)
            if not in_paren:
                # `int a = 0, b = 0, c;` -- the other declarators of the same statement
                rest = s[m.end():]
                for piece in split_top(rest)[1:]:
                    mm = re.match(r"\s*[\*&]*\s*(%s)\s*(?:=|$|\[)" % ident, piece)
                    if mm:
                        (local if d > 0 else decl0).add(mm.group(1))
        # --- assignments: root identifier of  name[...].member... OP=
        for m in re.finditer(r"(?<![\w.])(%s)((?:\s*\[[^\]]*\]|\s*(?:\.|->)\s*%s)*)\s*(<<=|>>=|\+=|-=|\*=|/=|\|=|&=|\^=|%%=|=)(?!=)" % (ident, ident), s):
            name, op = m.group(1), m.group(3)
            pre = s[:m.start()].rstrip()
            if op == "=" and pre.endswith(("=", "!", "<", ">")):
                continue
            if name in NOT_TYPES or name == "p":            # (p.scores[...] = ..., p.nz.value[...] = ...: stores to global memory through the argument block)
                continue
            assigned.add(name)
            if name.startswith("s_") and "[" in m.group(2):
                lds.add(name)
        for m in re.finditer(r"(?:\+\+|--)\s*(%s)|(%s)\s*(?:\+\+|--)" % (ident, ident), s):
            assigned.add(m.group(1) or m.group(2))
        # --- LDS atomics:  __hip_atomic_fetch_xxx(&s_name[...] ...  /  (cast)(&s_name[...])
        for m in re.finditer(r"&\s*(s_\w+)\s*\[", s):
            if "atomic" in s:
                lds.add(m.group(1)); assigned.add(m.group(1))
        # --- helpers that write through reference parameters
        for fn, pos in BY_REF.items():
            for m in re.finditer(r"\b%s\s*\(" % fn, s):
                args = split_top(s[m.end():].rsplit(")", 1)[0])
                for i in pos:
                    if i < len(args):
                        mm = re.match(r"\s*(%s)" % ident, args[i])
                        if mm:
                            assigned.add(mm.group(1))
    return dict(decl0=decl0, local=local, assigned=assigned, used=used, lds=lds)


def main(argv):
    global CSRC
    if len(argv) >= 2 and argv[0] == "--csrc":              # (the test of this tool points it at a doctored copy)
        CSRC, argv = argv[1], argv[2:]
    files = {ph: os.path.join(CSRC, f"bsx_step_phase_{ph}.inl") for ph in PHASES}
    def expand(path, depth=0):
        """A phase file's text with the part-files it includes (bsx_step_phase_shot_entry.inl, bsx_step_phase_respawn.inl) in place of the
        #include lines, as the compiler sees it: a part is text of the phase that includes it (a part included twice -- in place, and at the
        per-call two-wave kernel's deferred position -- counts for both includers)."""
        text = open(path).read()
        if depth > 2:
            return text
        return re.sub(r'^#include "(bsx_step_phase_\w+\.inl)"[^\n]*$', lambda m: expand(os.path.join(CSRC, m.group(1)), depth + 1), text, flags=re.M)
    raw = {ph: expand(f) for ph, f in files.items()}
    sc = {ph: scan(raw[ph]) for ph in PHASES}
    kernel = open(os.path.join(CSRC, "bsx_step_kernel.h")).read()
    after = strip(kernel.split('#include "bsx_step_phase_stores.inl"', 1)[1])
    used_after_kernel = set(re.findall(r"\b[A-Za-z_]\w*\b", after))
    errors = []
    derived = {}
    for i, ph in enumerate(PHASES):
        s = sc[ph]
        later_used = set().union(*[sc[q]["used"] for q in PHASES[i + 1:]], used_after_kernel) if True else set()
        earlier_exports = set().union(*[derived[q]["exports"] for q in PHASES[:i]]) if i else set()
        exports = {n for n in s["decl0"] if n in later_used}
        outside_written = {n for n in s["assigned"] if n not in s["decl0"] and n not in s["local"] and not n.startswith("s_")}
        reads = {n for n in earlier_exports if n in s["used"] and n not in s["decl0"]}
        derived[ph] = dict(exports=exports, writes=outside_written, reads=reads - outside_written, lds=s["lds"])
    if argv and argv[0] == "--print":
        for ph in PHASES:
            d = derived[ph]
            print(f"== {ph}")
            for tag in ("reads", "writes", "exports", "lds"):
                print(f"// @{tag:8s}" + " ".join(sorted(d[tag])))
        return 0
    for ph in PHASES:
        have, want = header_lists(raw[ph]), derived[ph]
        for tag in ("reads", "writes", "exports", "lds"):
            h, w = set(have[tag]), set(want[tag])
            if tag == "reads":
                w = w | (h & want["writes"])                 # a name both read and written may be listed under both
            if h != w:
                if w - h:
                    errors.append(f"bsx_step_phase_{ph}.inl: @{tag} misses {sorted(w - h)} (the text {'assigns' if tag in ('writes', 'lds') else 'uses / declares'} them)")
                if h - w:
                    errors.append(f"bsx_step_phase_{ph}.inl: @{tag} lists {sorted(h - w)} but the text does not bear it out")
    for e in errors:
        print(e)
    print(f"{len(PHASES)} phase files checked, {len(errors)} violation(s)")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
