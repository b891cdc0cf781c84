"""Minimal WebAssembly binary tooling for the pin against the reference's compiled module (docs/bonnie-engine.wasm):
section / name-section parser and an export adder.  Own code; nothing of the reference is reproduced here."""


def leb(b, p):
    r = s = 0
    while True:
        x = b[p]; p += 1
        r |= (x & 0x7F) << s; s += 7
        if not x & 0x80:
            return r, p


def uleb(v):
    out = bytearray()
    while True:
        x = v & 0x7F; v >>= 7
        out.append(x | (0x80 if v else 0))
        if not v:
            return bytes(out)


def sections(b):
    assert b[:4] == b"\0asm" and b[4:8] == b"\1\0\0\0"
    p, out = 8, []
    while p < len(b):
        sid = b[p]; hdr = p; p += 1
        n, p = leb(b, p)
        out.append((sid, hdr, p, n)); p += n
    return out


def function_names(b):
    """index -> name from the custom "name" section (subsection 1)."""
    names = {}
    for sid, _hdr, p, n in sections(b):
        if sid != 0:
            continue
        e = p + n
        l, q = leb(b, p)
        if b[q:q + l] != b"name":
            continue
        q += l
        while q < e:
            sub = b[q]; q += 1
            sl, q = leb(b, q); se = q + sl
            if sub == 1:
                c, q = leb(b, q)
                for _ in range(c):
                    idx, q = leb(b, q); l, q = leb(b, q)
                    names[idx] = b[q:q + l].decode(errors="replace"); q += l
            q = se
    return names


def add_exports(b, wanted):
    """wanted: export name -> substring of the (mangled) function name ("=name": the whole name).  Returns (new module bytes, export name -> function index).
    Only the export section is rewritten; every code and data byte stays as it is."""
    names = function_names(b)
    idx = {}
    for key, sub in wanted.items():
        hits = [i for i, n in names.items() if (n == sub[1:] if sub.startswith("=") else sub in n)]
        if len(hits) != 1:
            raise ValueError(f"{sub}: {len(hits)} functions match")
        idx[key] = hits[0]
    (sid, hdr, p, n), = [s for s in sections(b) if s[0] == 7]
    cnt, q = leb(b, p)
    extra = b"".join(uleb(len(k)) + k.encode() + b"\x00" + uleb(i) for k, i in idx.items())
    new = uleb(cnt + len(idx)) + b[q:p + n] + extra
    return b[:hdr] + b"\x07" + uleb(len(new)) + new + b[p + n:], idx
