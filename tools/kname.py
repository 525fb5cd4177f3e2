"""Canonical kernel names: rocprofv3's Kernel_Name / Name column (sometimes still mangled: hipcc's demangler does not know DF16_) ->
the form libcvmi355's cvmi_last_kernel() tags use, e.g. "tok_linear_kernel<576, 1, false, true, true, false, true, false>"."""
import functools
import re
import subprocess

CXXFILT = "c++filt"          # GNU binutils: does not know DF16_ / DF16b, so they are swapped for builtin codes it does know first


@functools.lru_cache(maxsize=None)
def canon(name):
    n = name.strip()
    if n.startswith("_Z"):
        try:
            m = n.replace("DF16b", "Ds").replace("DF16_", "Dh")            # builtin type codes: no substitution index moves
            d = subprocess.run([CXXFILT, m], capture_output=True, text=True, timeout=10).stdout.strip()
            if d and not d.startswith("_Z"):
                n = d.replace("char16_t", "__bf16").replace("half", "_Float16")
        except (OSError, subprocess.SubprocessError):
            pass
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void\s+", "", n)
    depth, out = 0, []
    for ch in n:                                   # cut the parameter list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()
