"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol
include/psdr.h declares; error behaviour without a device; derived parameters and the
roofline accounting of bench.py against SURVEY Appendix A / 8d."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import HAVE_GPU, ROOT


def _mkdtemp():
    """a scratch directory that does not outlive the test session"""
    import atexit
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="psdr_test_")
    atexit.register(shutil.rmtree, d, ignore_errors=True)
    return d


def header_functions():
    txt = open(os.path.join(ROOT, "include", "psdr.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(psdr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from phantomsdr_amd import _lib
    names = header_functions()
    assert len(names) >= 40
    L = C.CDLL(os.path.join(ROOT, "phantomsdr_amd", "libpsdr_hip.so"))
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/psdr.h but not exported"
    bound = {s[0] for s in _lib.SYMBOLS}
    assert set(names) <= bound | {"psdr_debug_trace"}, set(names) - bound
    assert _lib.load().psdr_version().startswith(b"phantomsdr_amd")


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-device error path")
def test_create_without_device_fails_loudly():
    import phantomsdr_amd as p
    with pytest.raises(p.PsdrError) as e:
        p.Context(1 << 16, False, 7)
    assert e.value.code == -2 and "No HIP devices" in str(e.value)


def test_config_struct_matches_header():
    from phantomsdr_amd._lib import psdr_config
    txt = open(os.path.join(ROOT, "include", "psdr.h")).read()
    body = txt[txt.index("typedef struct psdr_config {"): txt.index("} psdr_config;")]
    fields = re.findall(r"^\s+(?:u?int32_t)\s+([a-z_]+);", body, flags=re.M)
    assert fields == [f[0] for f in psdr_config._fields_]
    assert C.sizeof(psdr_config) == 4 * len(fields)


def test_option_and_fetch_constants_of_the_python_mirror_match_the_header():
    """psdr_set_option's knobs and psdr_fetch_begin's bits are plain integers in include/psdr.h; phantomsdr_amd.core carries
    copies - a renumbered enum would silently set another option."""
    from phantomsdr_amd.core import Context
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "psdr.h")).read(), flags=re.S)
    enum = dict((k, int(v)) for k, v in re.findall(r"(PSDR_OPT_[A-Z0-9_]+)\s*=\s*(\d+)", txt))
    assert enum == {"PSDR_OPT_POST_CHAIN_STREAMS": Context.OPT_POST_CHAIN_STREAMS, "PSDR_OPT_POST_CHAIN_AGC": Context.OPT_POST_CHAIN_AGC,
                    "PSDR_OPT_POST_CHAIN_PCM16": Context.OPT_POST_CHAIN_PCM16}
    bits = dict((k, int(v)) for k, v in re.findall(r"#define\s+(PSDR_FETCH_(?:AUDIO|PCM|WATERFALL))\s+(\d+)u", txt))
    assert bits == {"PSDR_FETCH_AUDIO": Context.FETCH_AUDIO, "PSDR_FETCH_PCM": Context.FETCH_PCM, "PSDR_FETCH_WATERFALL": Context.FETCH_WATERFALL}


def test_no_oracle_in_product_path():
    """the product never imports/links the oracle (it is test infrastructure only)"""
    pkg = os.path.join(ROOT, "phantomsdr_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                for needle in ("import oracle", "from oracle", "liboracle", "psdr_oracle", "orc_", "_ref/"):
                    assert needle not in src, (os.path.join(dp, f), needle)


@pytest.mark.parametrize("sps,N,is_real,n,levels,skip", [
    (3_200_000, 1 << 16, False, 248, 7, 8),      # cfg1
    (35_000_000, 1 << 20, False, 360, 11, 6),    # cfg2/4
    (70_000_000, 1 << 21, True, 360, 11, 6),     # cfg3
    (70_000_000, 1 << 22, True, 720, 12, 2),     # cfg5
    (20_000_000, 1 << 20, False, 10068, 10, 2),  # shipped config.toml (audio_sps 192k, wf 2048)
])
def test_derived_params_match_survey_appendix_a(sps, N, is_real, n, levels, skip):
    from phantomsdr_amd.core import derived_params
    kw = dict(audio_sps=192000, waterfall_size=2048) if n == 10068 else {}
    p = derived_params(sps, N, is_real, **kw)
    assert (p["audio_fft_size"], p["downsample_levels"], p["skip_num"]) == (n, levels, skip)
    assert p["fft_result_size"] == (N // 2 if is_real else N)


def test_bench_roofline_accounting_matches_survey_8d():
    import bench
    from phantomsdr_amd.core import derived_params
    wl = bench.WORKLOADS["cfg2"]
    p = derived_params(wl["sps"], wl["fft_size"], wl["is_real"])
    cl = bench.make_clients(wl, p, seed=1)
    wf = bench.make_waterfalls(wl, p, seed=1)
    assert len(cl) == 16 and len(wf) == 4
    ab = bench.algorithmic_bytes_per_frame(wl, p, cl, wf)
    assert ab["input"] == 4 * (1 << 20) and ab["spectrum"] == 8 * (1 << 20)
    assert ab["pyramid"] == sum((1 << 20) >> i for i in range(11))
    assert abs(ab["total"] - 14.73e6) < 0.05e6       # SURVEY 8d table, cfg 2
    for mode, l, m, r in cl:                          # 3 kHz SSB slices: 89 bins
        assert r - l == 89 and 0 <= l < r < (1 << 20)
    wl3 = bench.WORKLOADS["cfg3"]
    p3 = derived_params(wl3["sps"], wl3["fft_size"], wl3["is_real"])
    ab3 = bench.algorithmic_bytes_per_frame(wl3, p3, bench.make_clients(wl3, p3, seed=1), [])
    assert abs(ab3["total"] - 14.82e6) < 0.12e6       # cfg 3


def test_client_sharding_plan():
    from phantomsdr_amd.distributed import assign_clients
    plan = assign_clients(256, 8)
    assert all(len(x) == 32 for x in plan)
    assert sorted(sum(plan, [])) == list(range(256))
    assert plan[3][:3] == [3, 11, 19]
    assert assign_clients(5, 2) == [[0, 2, 4], [1, 3]]


def _build_c_example(name="level1_demo"):
    import subprocess
    import tempfile
    d = _mkdtemp()
    out = os.path.join(d, name)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"),
                           "-L" + os.path.join(ROOT, "phantomsdr_amd"), "-lpsdr_hip", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "phantomsdr_amd"), "-o", out])
    return d, out


def test_c_example_compiles_and_links_against_the_header():
    """examples/level1_demo.c: plain C over include/psdr.h (no GPU needed to build it)."""
    import subprocess
    _, out = _build_c_example()
    r = subprocess.run([out], capture_output=True, text=True)
    if HAVE_GPU:
        assert r.returncode == 0 and "peak bin 1000" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 2 and "No HIP devices found" in r.stderr


def test_group_example_compiles_and_fails_loudly_without_a_gpu():
    """examples/group_demo.c: psdr_group_* from plain C, linked against libpsdr_hip.so only (RCCL is the library's business)"""
    import subprocess
    _, out = _build_c_example("group_demo")
    r = subprocess.run([out, "1", "0"], capture_output=True, text=True)
    if HAVE_GPU:
        assert r.returncode == 0 and "group demo ok" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 2 and "No HIP devices found" in r.stderr


@pytest.mark.gpu
def test_c_example_runs_on_the_gpu_and_matches_the_oracle():
    """the plain-C caller drives Level 1 like broadcast_server::fft_task (src/fft.cpp:17-30,61-98);
    its dumped spectrum (N + additional bins, wrap copy applied) and int8 pyramid are compared with
    the oracle on the dumped input."""
    import subprocess
    import numpy as np
    from oracle import oracle as O
    d, out = _build_c_example()
    r = subprocess.run([out, d], capture_output=True, text=True)
    assert r.returncode == 0 and "peak bin 1000" in r.stdout, r.stdout + r.stderr
    N, A, levels = 1 << 16, 248, 7
    halves = [np.fromfile(os.path.join(d, f"half{h}.bin"), np.complex64) for h in range(3)]
    fo = O.FFT(N, False, levels, 0, A)
    for f in range(2):
        fo.load(halves[f], halves[f + 1])
        fo.execute()
        Xg = np.fromfile(os.path.join(d, f"spec{f}.bin"), np.complex64)
        qg = np.fromfile(os.path.join(d, f"q{f}.bin"), np.int8)
        Xo = fo.output()
        assert Xg.size == N + A and np.abs(Xg - Xo).max() <= 1e-4 * np.abs(Xo).max()
        assert np.array_equal(Xg[N:], Xg[:A])                      # src/fft.cpp:96-97
        assert np.array_equal(qg, O.pyramid_from_spectrum(Xg, N, False, levels))
        dq = np.abs(qg.astype(np.int16) - fo.quantized().astype(np.int16))
        assert dq.max() <= 1 and (dq != 0).mean() <= 1e-3


def test_cpp_adapter_mirrors_the_reference_interface():
    """phantomsdr_amd/host/hip_fft.h overrides every pure virtual of class FFT (src/fft.h:33-63)"""
    txt = open(os.path.join(ROOT, "phantomsdr_amd", "host", "hip_fft.h")).read()
    for member in ("malloc(size_t", "free(float", "plan_c2c(direction", "plan_r2c(int", "load_real_input(float",
                   "load_complex_input(float", "execute()", "get_output_buffer()", "get_quantized_buffer()"):
        assert member in txt, member
    assert "class hipFFT : public FFT" in txt


@pytest.mark.skipif(not os.path.isfile("/root/reference/src/fft.h"), reason="reference tree absent")
def test_cpp_adapter_compiles_against_the_reference_header():
    """`class hipFFT : public FFT` is compiled (syntax + semantics, g++ -fsyntax-only) against the
    reference's REAL src/fft.h, instantiated through the base-class pointer the server holds
    (src/spectrumserver.h: std::unique_ptr<FFT> fft).  src/fft.h includes <fftw3.h>, which this image
    lacks; for this COMPILE CHECK ONLY a forwarding header to ROCm's FFTW-API declarations
    (<hipfft/hipfftw.h>) is generated in a temporary directory - nothing is built or linked from it."""
    import subprocess
    import tempfile
    d = _mkdtemp()
    with open(os.path.join(d, "fftw3.h"), "w") as f:
        f.write("#include <hipfft/hipfftw.h>\n")
    with open(os.path.join(d, "tu.cpp"), "w") as f:
        f.write('#include <memory>\n#include "hip_fft.h"\n'
                "std::unique_ptr<FFT> make(size_t n, int levels) {\n"
                "    std::unique_ptr<FFT> f = std::make_unique<hipFFT>(n, 1, levels, 0);\n"
                "    f->set_output_additional_size(248);\n"
                "    f->plan_c2c(FFT::FORWARD, 0);\n"
                "    float *a = f->malloc(n), *b = f->malloc(n);\n"
                "    f->load_complex_input(a, b);\n"
                "    f->execute();\n"
                "    (void)f->get_output_buffer();\n"
                "    (void)f->get_quantized_buffer();\n"
                "    f->free(a);\n    f->free(b);\n    return f;\n}\n")
    subprocess.check_call(["g++", "-std=c++20", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unknown-pragmas",
                           "-I" + d, "-I/root/reference/src", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "phantomsdr_amd", "host"),
                           os.path.join(d, "tu.cpp")])


@pytest.mark.skipif(not os.path.isfile("/root/reference/src/fft.h"), reason="reference tree absent")
def test_integration_patches_apply_to_the_reference_tree():
    """integration/level1.patch and level2.patch (the f1 artefact: every edited line of src/fft.h,
    src/spectrumserver.cpp, meson.build, src/fft.cpp, src/websocket.cpp, src/signal.*, src/waterfall.*) apply
    cleanly, one after the other, to a pristine copy of the reference files, add lines only, and are what
    tools/make_integration_patches.py generates from the tree."""
    import shutil
    import subprocess
    import tempfile
    d = _mkdtemp()
    shutil.copytree("/root/reference/src", os.path.join(d, "src"))
    shutil.copy("/root/reference/meson.build", d)
    for name in ("level1.patch", "level2.patch"):
        txt = open(os.path.join(ROOT, "integration", name)).read()
        assert not [ln for ln in txt.splitlines() if ln.startswith("-") and not ln.startswith("---")], "additive only"
        subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.join(ROOT, "integration", name)], cwd=d)
    patched = open(os.path.join(d, "src", "spectrumserver.cpp")).read()
    assert "std::make_unique<hipFFT>" in patched and "std::make_unique<HipFanout>" in patched
    assert "GPU_hipFFT" in open(os.path.join(d, "src", "fft.h")).read()
    assert "fft_task_hip();" in open(os.path.join(d, "src", "fft.cpp")).read()
    assert os.path.exists(os.path.join(d, "meson_options.txt"))
    # the patched src/fft.h still compiles with the adapter on top of it (same compile check as above)
    with open(os.path.join(d, "fftw3.h"), "w") as f:
        f.write("#include <hipfft/hipfftw.h>\n")
    with open(os.path.join(d, "tu.cpp"), "w") as f:
        f.write('#include "hip_fft.h"\n#include "hip_fanout.h"\nfft_accelerator a = GPU_hipFFT;\n')
    subprocess.check_call(["g++", "-std=c++20", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unknown-pragmas", "-I" + d,
                           "-I" + os.path.join(d, "src"), "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "phantomsdr_amd", "host"), os.path.join(d, "tu.cpp")])


def test_level2_host_class_compiles_standalone():
    """phantomsdr_amd/host/hip_fanout.h needs nothing but include/psdr.h"""
    import subprocess
    import tempfile
    d = _mkdtemp()
    with open(os.path.join(d, "tu.cpp"), "w") as f:
        f.write('#include "hip_fanout.h"\nint use(HipFanout &f) { return f.add_audio_client(); }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "phantomsdr_amd", "host"), os.path.join(d, "tu.cpp")])


def test_level2_templates_compile_and_run_against_a_mock_of_the_reference():
    """phantomsdr_amd/host/hip_level2.h carries the bodies of fft_task_hip / send_audio_hip / send_waterfall_hip
    (integration/src/fft_hip.cpp only instantiates them).  tests/level2_mock/mock_reference.h declares the reference's
    classes with exactly the members they have (src/signal.h:53-123, src/waterfall.h:7-33, src/client.h:83-118,
    src/spectrumserver.h:88-175): the templates must compile against those names with -Wall -Werror, and the run
    checks what reaches the encoders - audio labels l = 0, m = audio_mid, r = r - l (src/signal.cpp:104-105, 287),
    NaN-dropped and never-attached clients send nothing, the CPU post chain runs exactly when the GPU's did not,
    waterfall rows on every skip_num-th frame, slow sockets skipped, no users -> no frame."""
    import subprocess
    d = _mkdtemp()
    src = os.path.join(ROOT, "tests", "level2_mock")
    exe = os.path.join(d, "run_level2")
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-O1", "-pthread", "-I" + src,
                           "-I" + os.path.join(ROOT, "phantomsdr_amd", "host"), os.path.join(src, "run_level2.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, timeout=120)
    assert r.returncode == 0 and b"level2 ok" in r.stdout, r.stderr.decode()


def test_level2_templates_instantiate_with_the_real_fanout_class():
    """the same templates with HipFanout (hip_fanout.h) in place of the scripted stand-in: every call they make on a
    fan-out exists there with a matching signature (-fsyntax-only: nothing is linked)"""
    import subprocess
    src = os.path.join(ROOT, "tests", "level2_mock")
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I" + src,
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "phantomsdr_amd", "host"),
                           os.path.join(src, "real_fanout_syntax.cpp")])


def test_level2_server_source_uses_only_the_templates():
    """integration/src/fft_hip.cpp must stay a thin instantiation: no member access of its own that the mock test
    would not see"""
    txt = open(os.path.join(ROOT, "integration", "src", "fft_hip.cpp")).read()
    code = "\n".join(ln for ln in txt.splitlines() if not ln.lstrip().startswith("//"))
    for ident in ("audio_real", "audio_l", "audio_r", "encoder", "signal_slices", "waterfall_slices", "frame_num++"):
        assert ident not in code, ident
    assert code.count("psdr_level2::Access::") == 3


# ---- examples/stream_demo.c: stdin -> ingest ring -> FFT -> demodulation / waterfall -> the reference's packets ----
STREAM_ARGS = dict(log2n=16, fmt="s16", sps=2_048_000, audio_sps=12000, batch=5)


def _stream_demo_cmd(exe):
    N = 1 << STREAM_ARGS["log2n"]
    a = [exe, str(STREAM_ARGS["log2n"]), "0", STREAM_ARGS["fmt"], str(STREAM_ARGS["sps"]), str(STREAM_ARGS["audio_sps"]),
         str(STREAM_ARGS["batch"])]
    clients = [("USB", 20000, 20000.5, 20090), ("AM", 41000, 41080.0, 41160), ("FM", 9000, 9100.25, 9200), ("LSB", 50000, 50096.0, 50096)]
    for mode, l, m, r in clients:
        a += ["--audio", '{"cmd":"window","l":%d,"r":%d,"m":%r}' % (l, r, m), '{"cmd":"demodulation","demodulation":"%s"}' % mode]
    wins = [(0, N), (30000, 32048)]
    for l, r in wins:
        a += ["--waterfall", '{"cmd":"window","l":%d,"r":%d}' % (l, r)]
    return a, clients, wins


def test_stream_demo_compiles_and_fails_loudly_without_a_gpu():
    import subprocess
    _, exe = _build_c_example("stream_demo")
    cmd, _, _ = _stream_demo_cmd(exe)
    r = subprocess.run(cmd, input=b"", capture_output=True)
    if not HAVE_GPU:
        assert r.returncode == 2 and b"No HIP devices found" in r.stderr


@pytest.mark.gpu
def test_stream_demo_end_to_end_against_the_oracle():
    """raw cs16 on stdin through the ingest ring (13 half-frames: batches of 5, 5 and 2 frames, copies running
    ahead of the transforms), four audio clients and two waterfall clients configured with the reference's JSON
    command frames; stdout carries the hello frame, one audio CBOR packet per client and frame and one zstd-
    streamed waterfall CBOR packet per client and sent frame.  Every packet is decoded here (own CBOR decoder,
    libzstd's streaming API) and its payload compared with the oracle run on the same samples."""
    import ctypes as C
    import json
    import subprocess
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import check_fm, quantize_raw, synth_stream
    from oracle import oracle as O
    from test_wire_formats import cbor_decode
    from phantomsdr_amd.core import derived_params
    _, exe = _build_c_example("stream_demo")
    cmd, clients, wins = _stream_demo_cmd(exe)
    N, nfr = 1 << STREAM_ARGS["log2n"], 12
    x = synth_stream((nfr + 1) * (N // 2), False, seed=33, fft_size=N)
    raw = quantize_raw(x, "s16", False)
    r = subprocess.run(cmd, input=raw.tobytes(), capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    p = derived_params(STREAM_ARGS["sps"], N, False, STREAM_ARGS["audio_sps"], 1024)
    n, levels, skip = p["audio_fft_size"], p["downsample_levels"], p["skip_num"]
    assert ("%d frames of %d points" % (nfr, N)).encode() in r.stderr and ("n = %d, levels = %d, skip = %d" % (n, levels, skip)).encode() in r.stderr
    # split the record stream
    recs, b, i = [], r.stdout, 0
    while i < len(b):
        kind, cl, ln = chr(b[i]), int.from_bytes(b[i + 1:i + 5], "little"), int.from_bytes(b[i + 5:i + 9], "little")
        recs.append((kind, cl, b[i + 9:i + 9 + ln]))
        i += 9 + ln
    assert i == len(b)
    hello = json.loads(recs[0][2])
    assert recs[0][0] == "H" and hello["fft_size"] == N and hello["audio_max_fft"] == n and hello["defaults"]["l"] == clients[0][1]
    # oracle on the same samples
    conv = O.convert(raw, "s16").view(np.complex64).reshape(nfr + 1, N // 2)
    fo = O.FFT(N, False, levels, 0, n)
    ocl = []
    for mode, l, m, rr in clients:
        c = O.AudioClient(False, n, STREAM_ARGS["audio_sps"], N)
        c.set_audio_demodulation(mode)
        assert c.on_window_message(l, m, rr)
        ocl.append(c)
    want_audio, want_q = {}, []
    for f in range(nfr):
        fo.load(conv[f], conv[f + 1])
        fo.execute()
        spec = fo.output().copy()
        want_q.append(fo.quantized().copy())
        for ci, c in enumerate(ocl):
            a_o, p_o, _, dropped = c.send_audio(spec, f, fft=fo)
            assert not dropped
            want_audio[(ci, f)] = (a_o, p_o, c.mode, c.baseband()[: n // 2], c.bb_prev)
    # audio packets: one per client and frame, in frame order per client, labelled with the client's window
    seen = {}
    for kind, cl, body in recs:
        if kind != "A":
            continue
        d = cbor_decode(body)
        assert list(d) == ["data", "frame_num", "l", "m", "pwr", "r"]
        mode, l, m, rr = clients[cl]
        # labels as AudioClient::send_audio sends them (src/signal.cpp:104-105, 287): l = audio_l = l - l = 0,
        # m = audio_mid (absolute), r = audio_r = r - l
        assert (d["l"], d["m"], d["r"]) == (0, m, rr - l)
        f = d["frame_num"]
        assert f == seen.get(cl, -1) + 1
        seen[cl] = f
        a_g = np.frombuffer(d["data"], np.float32)
        a_o, p_o, omode, bb, bb_prev = want_audio[(cl, f)]
        assert a_g.size == n // 2
        assert abs(d["pwr"] - p_o) <= 1e-4 * max(abs(p_o), 1e-30) + 1e-30
        if omode == O.FM:
            check_fm(a_g, a_o, bb, bb_prev, f"client {cl} frame {f}")
        else:
            assert np.abs(a_g - a_o).max() <= 3e-4 * max(np.abs(a_o).max(), 1e-30) + 1e-9, (cl, f)
    assert seen == {ci: nfr - 1 for ci in range(len(clients))}
    # waterfall packets: every skip-th frame, through ONE zstd stream per client
    z = None
    try:
        z = C.CDLL("libzstd.so.1")
    except OSError:
        pass
    kinds = {k for k, _, _ in recs}
    assert ("Z" in kinds) == (z is not None) and ("Z" in kinds) != ("W" in kinds)

    class Buf(C.Structure):
        _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]
    if z is not None:
        z.ZSTD_createDStream.restype = C.c_void_p
        z.ZSTD_decompressStream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        z.ZSTD_decompressStream.restype = C.c_size_t
        z.ZSTD_freeDStream.argtypes = [C.c_void_p]
    streams, nsent = {}, {}
    for kind, cl, body in recs:
        if kind not in "WZ":
            continue
        if kind == "Z":
            ds = streams.setdefault(cl, z.ZSTD_createDStream())
            src = (C.c_uint8 * len(body)).from_buffer_copy(body)
            dst = (C.c_uint8 * (N + 256))()
            ib, ob = Buf(C.cast(src, C.c_void_p), len(body), 0), Buf(C.cast(dst, C.c_void_p), len(dst), 0)
            z.ZSTD_decompressStream(ds, C.byref(ob), C.byref(ib))
            assert ib.pos == len(body)          # a flushed packet decodes completely on arrival
            body = bytes(dst[: ob.pos])
        d = cbor_decode(body)
        assert list(d) == ["data", "frame_num", "l", "r"]
        f = d["frame_num"]
        assert f % skip == 0 and f == nsent.get(cl, -skip) + skip
        nsent[cl] = f
        # the level WaterfallClient::on_window_message picks for this window (src/waterfall.cpp:62-79)
        l0, r0 = wins[cl]
        best, lv, lf, rf, ll, rr = 2048.0, levels - 1, float(l0), float(r0), l0, r0
        for i2 in range(levels):
            if abs((rf - lf) - 1024.0) < best:
                best, lv, ll, rr = abs((rf - lf) - 1024.0), i2, int(round(lf)), int(round(rf))
            lf, rf = lf / 2, rf / 2
        assert (d["l"], d["r"]) == (ll << lv, rr << lv)
        row = np.frombuffer(d["data"], np.int8)
        off = sum(N >> t for t in range(lv))
        want = want_q[f][off + ll: off + rr]
        dq = np.abs(row.astype(np.int16) - want.astype(np.int16))
        assert row.size == rr - ll and dq.max() <= 1 and (dq != 0).mean() <= 5e-3, (cl, f, dq.max())
    last = ((nfr - 1) // skip) * skip
    assert nsent == {0: last, 1: last}
    for ds in streams.values():
        z.ZSTD_freeDStream(ds)
