"""The C-ABI library loads without a GPU, exports every symbol include/lgstep.h declares, and agrees with the ctypes
mirror (extended_legged_gym_amd/abi.py) on enum values and struct sizes.  No compute calls here."""
import ctypes
import os
import re

from extended_legged_gym_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "lgstep.h")).read()
LIB = os.path.join(ROOT, "extended_legged_gym_amd", "csrc", "liblgstep.so")


def parse_enum(name):
    body = re.search(r"enum\s+%s\s*\{(.*?)\};" % name, HEADER, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    vals, cur = {}, -1
    for item in body.split(","):
        item = item.strip()
        if not item:
            continue
        if "=" in item:
            k, v = [x.strip() for x in item.split("=")]
            cur = int(v, 0)
        else:
            k, cur = item, cur + 1
        vals[k] = cur
    return vals


def test_enums_match_the_header():
    t = parse_enum("lg_tensor_id")
    assert t.pop("LG_T_COUNT") == abi.LG_T_COUNT
    assert {k[len("LG_T_"):].lower(): v for k, v in t.items()} == abi.TENSOR_ID
    r = parse_enum("lg_reward_term")
    assert r.pop("LG_REW_COUNT") == len(abi.REWARD_TERMS)
    assert {k[len("LG_REW_"):].lower(): v for k, v in r.items()} == abi.REWARD_TERM_ID
    head = parse_enum("lg_rand_slot")            # the slots in front of the per-DOF draws; the rest follow from the DOF count
    assert head == {k: v for k, v in abi.RAND_SLOTS.items() if k in head} and set(head) == {"LG_RS_CMD_CB", "LG_RS_PUSH", "LG_RS_LEVEL", "LG_RS_DOF"}
    assert abi.rand_slots(12) == dict(head, LG_RS_ROOT_XY=20, LG_RS_ROOT_VEL=22, LG_RS_CMD_RESET=28, LG_RS_NOISE=32)
    assert abi.rand_slots(18) == dict(head, LG_RS_ROOT_XY=26, LG_RS_ROOT_VEL=28, LG_RS_CMD_RESET=34, LG_RS_NOISE=40)
    for macro in ("LG_RS_ROOT_XY_OF(dof)   (LG_RS_DOF + (dof))", "LG_RS_ROOT_VEL_OF(dof)  (LG_RS_ROOT_XY_OF(dof) + 2)",
                  "LG_RS_CMD_RESET_OF(dof) (LG_RS_ROOT_VEL_OF(dof) + 6)", "LG_RS_NOISE_OF(dof)     ((LG_RS_CMD_RESET_OF(dof) + 3 + 3) & ~3)",
                  "LG_NUM_PROPRIO_OF(dof)  (12 + 3 * (dof))"):
        assert "#define " + macro in HEADER, macro
    assert (abi.num_proprio(12), abi.num_proprio(18)) == (48, 66)
    d = parse_enum("lg_dtype")
    assert (d["LG_F32"], d["LG_I64"], d["LG_U8"], d["LG_I16"], d["LG_I32"], d["LG_F64"]) == (
        abi.LG_F32, abi.LG_I64, abi.LG_U8, abi.LG_I16, abi.LG_I32, abi.LG_F64)
    c = parse_enum("lg_control")
    assert c == dict(LG_CTRL_P=abi.LG_CTRL_P, LG_CTRL_V=abi.LG_CTRL_V, LG_CTRL_T=abi.LG_CTRL_T,
                     LG_CTRL_ACTUATOR_NET=abi.LG_CTRL_ACTUATOR_NET)
    for macro, val in [("LG_MAX_CP", abi.LG_MAX_CP), ("LG_MAX_REWARD_TERMS", abi.LG_MAX_REWARD_TERMS),
                       ("LG_LSTM_NPARAM", abi.LG_LSTM_NPARAM), ("LG_ABI_VERSION", abi.LG_ABI_VERSION),
                       ("LG_MAX_INDEX_LIST", abi.LG_MAX_INDEX_LIST), ("LG_MAX_LEGS", abi.LG_MAX_LEGS), ("LG_MAX_BODIES", abi.LG_MAX_BODIES)]:
        assert int(re.search(r"#define\s+%s\s+(\d+)" % macro, HEADER).group(1)) == val


def test_library_exports_every_declared_symbol_and_struct_sizes_agree():
    declared = sorted(set(re.findall(r"\b(lg_[a-z_]+)\s*\(", HEADER)))
    assert sorted(abi.PRODUCT_SYMBOLS) == declared
    lib = ctypes.CDLL(LIB)
    for sym in declared:
        assert hasattr(lib, sym), sym
    abi.declare_product(lib)
    sizes = (ctypes.c_int32 * 4)()
    lib.lg_abi_sizes(sizes)
    assert list(sizes) == [abi.LG_ABI_VERSION, ctypes.sizeof(abi.lg_config), ctypes.sizeof(abi.lg_robot_model),
                           ctypes.sizeof(abi.lg_terrain)]


def test_policy_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "lgpolicy.h")).read()
    declared = sorted(set(re.findall(r"\b(lg_[a-z_]+)\s*\(", hdr)))
    assert sorted(abi.POLICY_SYMBOLS) == declared
    lib = ctypes.CDLL(LIB)
    for sym in declared:
        assert hasattr(lib, sym), sym
    abi.declare_policy(lib)
    body = re.search(r"enum\s+lg_activation\s*\{(.*?)\};", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    vals = {k.strip(): int(v) for k, v in (item.split("=") for item in body.split(",") if "=" in item)}
    assert {k[len("LG_ACT_"):].lower(): v for k, v in vals.items()} == abi.ACTIVATIONS


def test_product_has_no_path_through_the_oracle():
    """The package must never import or link the checker."""
    pkg = os.path.join(ROOT, "extended_legged_gym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "liblg_oracle", "lgo_", "oracle_lib", "lg_oracle"):
                    assert needle not in txt, f"{f} reaches into the oracle ({needle})"


def test_layer_parameter_structs_mirror_the_header(tmp_path):
    """The ctypes mirrors of the device layers' parameter blocks (`lg_pose_params`, `lg_foottrack_params`, `lg_foottrack_state`: passed by value into kernels)
    against `sizeof` / `offsetof` as the C compiler lays the header's structs out."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lgstep.h"\nint main(void) {\n'
                   '  printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(lg_pose_params), sizeof(lg_foottrack_params), sizeof(lg_foottrack_state),\n'
                   '         offsetof(lg_foottrack_params, scales), offsetof(lg_foottrack_params, feet_indices), offsetof(lg_foottrack_state, fw_timer));\n  return 0;\n}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(abi.lg_pose_params), C.sizeof(abi.lg_foottrack_params), C.sizeof(abi.lg_foottrack_state),
            abi.lg_foottrack_params.scales.offset, abi.lg_foottrack_params.feet_indices.offset, abi.lg_foottrack_state.fw_timer.offset]
    assert got == want
