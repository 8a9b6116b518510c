"""The committed JNI binding (jni/MleaseHip.java + jni/mlease_jni.c) cannot be built here (no JDK); these tests keep it from
rotting: the C glue must type-check against the C-ABI header and a stub jni.h, every entry point of include/mlease_admm.h
must be bound, and every native method of the Java class must have its Java_... function (and vice versa)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "jni", "mlease_jni.c")
JAVA = os.path.join(ROOT, "jni", "MleaseHip.java")
HEADER = os.path.join(ROOT, "include", "mlease_admm.h")


def test_glue_type_checks_against_the_abi_header():
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
                        "-I", os.path.join(ROOT, "tests", "jni_stub"), "-I", os.path.join(ROOT, "include"), GLUE],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_every_abi_entry_point_is_bound_and_natives_match():
    header = open(HEADER).read()
    glue = open(GLUE).read()
    java = open(JAVA).read()
    entry_points = set(re.findall(r"^\s*(?:const char \*|int )\s*(mlx_[a-z_0-9]+)\s*\(", header, re.M))
    assert len(entry_points) >= 29, sorted(entry_points)
    missing = [f for f in sorted(entry_points) if not re.search(r"\b%s\s*\(" % f, glue)]
    assert not missing, "C-ABI entry points the JNI glue never calls: %s" % missing
    natives = set(re.findall(r"\bnative\s+[\w\[\].]+\s+(\w+)\s*\(", java))
    cfuncs = set(re.findall(r"^JFN\([\w ]+,\s*(\w+)\)", glue, re.M))          # (the #define line itself does not start a line with JFN)
    assert natives == cfuncs, "natives without glue: %s ; glue without native: %s" % (sorted(natives - cfuncs), sorted(cfuncs - natives))
    # the error mapping of jobs/RegressionAdmmTrain.java:713-716 and utils/LinearModelUtils.java:80-83
    assert '"Model fitting error!"' in glue and '"Some models failed!"' in glue
