"""The C ABI as three documents that must say the same thing: include/cara_hip.h (the contract), cara_amd/_lib.py
(the ctypes mirror the product uses) and the binding snippet of INTEGRATION.md (what a maintainer copies).  A field
missing at the END of a mirror hands the library a struct that is too short -- it then reads the missing member from
whatever follows -- so names, order, C types and sizeof are compared, not just the symbol list.  CPU only."""
import ctypes as C
import os
import re

from cara_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "cara_hip.h")).read()


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def header_struct(name):
    """[(field, ctype-class)] of `typedef struct { ... } name;` in declaration order."""
    m = re.search(r"typedef\s+struct\s*\{([^{}]*)\}\s*" + name + r"\s*;", _strip_comments(HDR), flags=re.S)
    assert m, name
    out = []
    for decl in m.group(1).split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        base, rest = re.match(r"((?:const\s+)?(?:unsigned\s+)?(?:long\s+long|size_t|int|float|void|unsigned|double)\s*)(.*)", decl).groups()
        base = base.replace("const", "").strip()
        for item in rest.split(","):
            item = item.strip()
            is_ptr = item.startswith("*") or base.endswith("*")
            field = item.lstrip("* ").strip()
            if is_ptr:
                kind = C.c_void_p
            else:
                kind = {"int": C.c_int, "float": C.c_float, "size_t": C.c_size_t, "long long": C.c_longlong, "unsigned": C.c_uint}[base]
            out.append((field, kind))
    return out


MIRRORS = {
    "cara_gemm_args": _lib.GemmArgs, "cara_geom": _lib.Geom, "cara_cp": _lib.CpPtrs, "cara_pack_layout": _lib.PackLayout,
    "cara_layer_grads": _lib.LayerGrads, "cara_vit_weights": _lib.VitWeights, "cara_vit_shape": _lib.VitShape,
    "cara_ts_reduce": _lib.TsReduce, "cara_linear": _lib.Linear,
}


def test_every_ctypes_mirror_matches_the_header_field_by_field():
    for cname, mirror in MIRRORS.items():
        want = header_struct(cname)
        got = [(n, t) for n, t in mirror._fields_]
        assert [n for n, _ in got] == [n for n, _ in want], (cname, [n for n, _ in got], [n for n, _ in want])
        for (n, tg), (_, tw) in zip(got, want):
            assert C.sizeof(tg) == C.sizeof(tw) and (tg is C.c_void_p) == (tw is C.c_void_p), (cname, n, tg, tw)


def test_struct_ids_of_the_header_follow_the_mirror_table():
    m = re.search(r"CARA_STRUCT_GEMM_ARGS\s*=\s*0(.*?)CARA_STRUCT_COUNT", _strip_comments(HDR), flags=re.S)
    ids = ["CARA_STRUCT_GEMM_ARGS"] + re.findall(r"CARA_STRUCT_[A-Z_]+", m.group(1))
    # (cara_adamw_args holds arrays and a nested struct, which the field parser above does not read: its element struct is
    # compared field by field below, the whole by sizeof -- lib() and test_library_reports_the_same_struct_sizes)
    names = ["CARA_STRUCT_" + n[len("cara_"):].upper() for n in MIRRORS] + ["CARA_STRUCT_ADAMW_ARGS"]
    assert ids == names, (ids, names)
    assert tuple(MIRRORS.values()) + (_lib.AdamWArgs,) == _lib.STRUCT_MIRRORS


def test_adamw_tensor_entry_matches_the_header():
    want = header_struct("cara_adamw_tensor")
    got = list(_lib.AdamWTensor._fields_)
    assert [n for n, _ in got] == [n for n, _ in want]
    for (n, tg), (_, tw) in zip(got, want):
        assert C.sizeof(tg) == C.sizeof(tw) and (tg is C.c_void_p) == (tw is C.c_void_p), (n, tg, tw)
    m = re.search(r"#define\s+CARA_ADAMW_MAX_TENSORS\s+(\d+).*?#define\s+CARA_ADAMW_MAX_GROUPS\s+(\d+)", HDR, flags=re.S)
    assert (int(m.group(1)), int(m.group(2))) == (_lib.ADAMW_MAX_TENSORS, _lib.ADAMW_MAX_GROUPS)


def test_library_reports_the_same_struct_sizes():
    lib = _lib.lib()   # (lib() itself raises on a mismatch; asserted here once more, explicitly)
    for which, mirror in enumerate(_lib.STRUCT_MIRRORS):
        assert int(lib.cara_sizeof_struct(which)) == C.sizeof(mirror), mirror.__name__
    assert int(lib.cara_sizeof_gemm_args()) == C.sizeof(_lib.GemmArgs)
    assert int(lib.cara_sizeof_struct(len(_lib.STRUCT_MIRRORS))) == 0


def test_integration_md_snippet_is_the_header_struct():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class GemmArgs\(C\.Structure\):.*?_fields_\s*=\s*\[(.*?)\]\n", md, flags=re.S)
    assert m, "INTEGRATION.md lost its GemmArgs snippet"
    body = re.sub(r"#.*", "", m.group(1))
    got = re.findall(r'\("(\w+)",\s*C\.(\w+)\)', body)
    want = header_struct("cara_gemm_args")
    assert [n for n, _ in got] == [n for n, _ in want], ([n for n, _ in got], [n for n, _ in want])
    for (n, t), (_, tw) in zip(got, want):
        assert getattr(C, t) is tw or C.sizeof(getattr(C, t)) == C.sizeof(tw) and tw is not C.c_void_p, (n, t, tw)
    # and the snippet, built as written, has the library's size
    fields = [(n, getattr(C, t)) for n, t in got]
    snippet = type("GemmArgsDoc", (C.Structure,), {"_fields_": fields})
    assert C.sizeof(snippet) == int(_lib.lib().cara_sizeof_gemm_args())
