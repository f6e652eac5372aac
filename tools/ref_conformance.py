"""Compile the reference's driver sources UNCHANGED against this library's headers.

The reference text is read where it lies (/root/reference); nothing of it is copied into the repo.  A translation
unit is handed to hipcc as it is -- its quote-includes of sibling reference files (pseudoXGCmTypes.hpp,
ellipticalPush.hpp, gyroScatter.hpp, perfTypes.hpp, Distribute.h) resolve next to it, every library header it names
(<particle_structs.hpp>, "pumipic_adjacency.hpp", <Omega_h_mesh.hpp>, <Kokkos_Core.hpp>, ...) resolves to
pumi-pic_amd/include[/compat].  No -I points into the reference tree.

  python tools/ref_conformance.py            # syntax check of every unit, N-of-M report
  python tools/ref_conformance.py --build    # also link executables into tests/_refdrivers/ (git-ignored; they
                                             # travel to the GPU box like the library's .so)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PUMIPIC_REFERENCE", "/root/reference")
INC = os.path.join(ROOT, "pumi-pic_amd", "include")
OUT = os.path.join(ROOT, "tests", "_refdrivers")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -DNDEBUG: a Release build, as the reference's performance runs are (test/gyroScatter.hpp:68 asserts that every ring
# point of the gyro map is found within 100 walk iterations, which a mesh with thin elements does not grant)
FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O2", "-DNDEBUG", "-x", "hip", "-ffp-contract=off", "-fno-fast-math",
         "-DFP64", "-DPP_USE_HIP", "-DPP_USE_GPU", "-Wno-unused-result", "-Wno-unused-value", "-Wno-unused-variable",
         "-I", os.path.join(INC, "compat"), "-I", INC]

# executable -> the reference translation units it is made of (reference-relative paths)
UNITS = {
    "pseudoXGCm": ["test/pseudoXGCm.cpp"],
    "pseudoPushAndSearch": ["test/pseudoPushAndSearch.cpp"],
    "ps_combo160": ["performance_tests/ps_combo160.cpp", "particle_structs/test/Distribute.cpp"],
    "ps_combo264": ["performance_tests/ps_combo264.cpp", "particle_structs/test/Distribute.cpp"],
    # the reference's OWN tests of the particle-structure API (particle_structs/test/testing.cmake:1-33): the particle
    # file writer, then test_structure -- counts, parallel_for, getPIDs, five rebuild scenarios, two migrations, copy to
    # the host and back, SubSegment, migrate-to-empty-and-refill -- on SCS (two parameter sets) and CSR, 1 and 4 ranks
    "write_particles": ["particle_structs/test/write_particle_file.cpp", "particle_structs/test/Distribute.cpp"],
    "test_structure": ["particle_structs/test/test_structure.cpp"],
    "typeTest": ["particle_structs/test/typeTest.cpp", "particle_structs/test/Distribute.cpp"],
    "initParticles": ["particle_structs/test/initParticles.cpp", "particle_structs/test/Distribute.cpp"],
    "buildSCSTest": ["particle_structs/test/buildSCSTest.cpp", "particle_structs/test/Distribute.cpp"],
    "lambdaTest": ["particle_structs/test/lambdaTest.cpp", "particle_structs/test/Distribute.cpp"],
    "test_scs_padding": ["particle_structs/test/scs_padding.cpp", "particle_structs/test/Distribute.cpp"],
    # the reference's own 2-D adjacency-search test (test/CMakeLists.txt search2d): 14 one-particle walks on the 8-triangle
    # plate, each closed by a device-side assert on the destination element -- built WITH asserts (see ASSERTS_ON)
    "search2d": ["test/search2d.cpp"],
    # the reference's search test proper (testing.cmake: test_adj_2d / test_adj_3d): 100 and 10^6 particles placed
    # inside elements and on vertices / edges / faces, pushed, searched by barycentric walk and by intersection with
    # the wall, each result judged by the reference's own check_initial_parents / wall-intersection checks
    "test_adj": ["test/test_adj.cpp"],
    # moller_trumbore_test (testing.cmake): ray against segment on the faces of one tet
    "moller_trumbore_test": ["test/moller_trumbore_line_tri_test.cpp"],
    # pseudoXGCm_scatter (testing.cmake): the gyro scatter of one particle on the 8-triangle plate, known vertex sums
    "pseudoXGCm_scatter": ["test/pseudoXGCm_scatter.cpp"],
    # barycentric (testing.cmake: barycentric_3): find_barycentric_tet known answers -- through src/unit_tests.hpp, see
    # EXTRA_FLAGS
    "barycentric": ["test/test_barycentric.cpp"],
    # the rows either side of the path: PICparts from an Input (input_construct_cube, 4 ranks) and the
    # particle balancer on an array and on a structure (lb_r1 / lb_r4)
    "input_construct": ["test/test_input_construct.cpp"],
    "test_lb": ["test/test_lb.cpp"],
    # comm arrays of the parts (comm_array_pisces / comm_array_2d_box, 4 ranks): SUM over a full buffer, MIN of the
    # owners, 1/n contributions summed to 1, a 3-component element array
    "comm_array": ["test/test_comm_array.cpp"],
    # parts with ghost / safe layers carry the full mesh's element ids (ptn_loading_cube, 2 and 4 ranks); an Input read
    # from a .ptn file with FULL buffers reproduces the serial mesh (full_mesh_pisces, 4 ranks)
    "ptn_loading": ["test/test_ptn_loading.cpp"],
    "full_mesh": ["test/test_full_mesh.cpp"],
    # pumipic::write then pumipic::read: the part read back equals the part written, array by array (file_rw_cube_4 and
    # the 1-rank form; the checks are asserts) -- through this library's own container, not .osh / .ppm
    "file_rw": ["test/test_file.cpp"],
}
# units whose checks are assert()s: compiled without -DNDEBUG so that a wrong destination element aborts the program
ASSERTS_ON = {"search2d", "test_adj", "pseudoXGCm_scatter", "file_rw"}
# test/test_barycentric.cpp includes "unit_tests.hpp", a header of test functions that the reference keeps in src/.
# It is read where it lies through -idirafter (searched AFTER this library's include directories, so nothing else
# resolves there); the one library header it names next to itself ("pumipic_adjacency.hpp", quote form: its own
# directory first) (and the .tpp that file appends) is emptied by predefining the reference's two include guards, and this library's header of that name is
# put in front with -include.  No token of a reference library header is compiled.
EXTRA_FLAGS = {"barycentric": ["-DPUMIPIC_ADJACENCY_HPP", "-DPUMIPIC_ADJACENCY_NEW_HPP", "-include", os.path.join(INC, "pumipic_adjacency.hpp"),
                               "-idirafter", os.path.join(REF, "src")]}


def have_reference():
    return os.path.isdir(os.path.join(REF, "test"))


def _extra(rel):
    return [f for name, fl in EXTRA_FLAGS.items() if rel in UNITS[name] for f in fl]


def syntax_check(rel):
    """(ok, diagnostics) of `hipcc -fsyntax-only` on the unchanged reference file `rel` (host and device passes)."""
    cmd = [HIPCC] + FLAGS + _extra(rel) + ["-fsyntax-only", os.path.join(REF, rel)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return r.returncode == 0, r.stderr


def build(name):
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, name)
    srcs = [os.path.join(REF, rel) for rel in UNITS[name]]
    flags = [f for f in FLAGS if not (name in ASSERTS_ON and f == "-DNDEBUG")]
    cmd = [HIPCC] + flags + EXTRA_FLAGS.get(name, []) + srcs + ["-o", exe, "-L", os.path.join(ROOT, "pumi-pic_amd"), "-lpumipic_hip",
                                   "-Wl,-rpath,$ORIGIN/../../pumi-pic_amd"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return r.returncode == 0, r.stderr, exe


def build_all(workers=4):
    """every executable of UNITS, a few hipcc runs side by side -> {name: (ok, diagnostics, path)}"""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return dict(zip(UNITS, ex.map(build, UNITS)))


def main():
    if not have_reference():
        print("no reference tree at", REF)
        return 0
    rels = sorted({rel for v in UNITS.values() for rel in v})
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    if only:
        rels = [r for r in rels if any(o in r for o in only)]
    ok_n = 0
    for rel in rels:
        ok, err = syntax_check(rel)
        ok_n += ok
        print("%-45s %s" % (rel, "compiles unchanged" if ok else "FAILS"))
        if not ok:
            errs = [l for l in err.splitlines() if "error" in l]
            print("\n".join("    " + l for l in errs[:int(os.environ.get("NERR", "25"))]))
    print("%d of %d reference translation units compile unchanged" % (ok_n, len(rels)))
    bad = len(rels) - ok_n
    if "--build" in sys.argv:
        for name, (ok, err, exe) in build_all().items():
            print("%-24s %s" % (name, exe if ok else "LINK FAILS\n" + err[-2000:]))
            bad += not ok
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
