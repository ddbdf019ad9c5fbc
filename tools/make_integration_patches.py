#!/usr/bin/env python
"""Regenerates integration/level1.patch and integration/level2.patch from the reference tree.
The patches are ADDITIVE (-U0, no context, no removed lines): they carry only this repository's lines
and the line numbers of the reference files they go into - no reference text is copied.
  level1: `class hipFFT` as a fourth FFT back-end (src/fft.h:26-31, src/spectrumserver.cpp:173-213, meson.build:87-109)
  level2: the per-frame fan-out on the GPU (src/fft.cpp:10, src/websocket.cpp:129-200, src/signal.*, src/waterfall.*)
usage: tools/make_integration_patches.py [/root/reference]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


def build(edits, newfiles, out):
    tmp = tempfile.mkdtemp()
    for f, _ in edits:
        for side in "ab":
            os.makedirs(os.path.dirname(os.path.join(tmp, side, f)), exist_ok=True)
            if not os.path.exists(os.path.join(tmp, side, f)):
                shutil.copy(os.path.join(REF, f), os.path.join(tmp, side, f))
    for f, fn in edits:
        p = os.path.join(tmp, "b", f)
        lines = open(p).read().split("\n")
        open(p, "w").write("\n".join(fn(lines)))
    for f, txt in newfiles.items():
        os.makedirs(os.path.dirname(os.path.join(tmp, "b", f)) or tmp, exist_ok=True)
        open(os.path.join(tmp, "b", f), "w").write(txt)
    o = subprocess.run(["diff", "-U0", "-r", "-N", "a", "b"], cwd=tmp, capture_output=True, text=True).stdout
    o = re.sub(r"^(---|\+\+\+) (\S+)\t.*$", r"\1 \2", o, flags=re.M)  # no timestamps
    assert not [ln for ln in o.split("\n") if ln.startswith("-") and not ln.startswith("---")], "must be additive"
    open(out, "w").write(o)
    return o


def after(lines, needle, new, offset=1, nth=0):
    i = [k for k, ln in enumerate(lines) if needle in ln][nth]
    return lines[:i + offset] + new + lines[i + offset:]


H = lambda body: ["#ifdef PSDR_HIP"] + body + ["#endif"]


# ------------------------------------------------------------------------------------ level 1
e1 = [
    ("src/fft.h", lambda L: after(L, "CPU_mklFFT,", [
        "    GPU_hipFFT,  // MI355X: hand-written HIP kernels behind libpsdr_hip.so (hip_fft.h)"])),
    ("src/spectrumserver.cpp", lambda L: after(after(after(
        L, '#include "spectrumserver.h"', H(['#include "hip_fft.h"  // class hipFFT : public FFT, forwards to libpsdr_hip.so'])),
        'std::cout << "Using MKL" << std::endl;', [
            '    } else if (accelerator_str == "hip") {',
            "        accelerator = GPU_hipFFT;",
            '        std::cout << "Using HIP (gfx950)" << std::endl;']),
        'throw "MKL support is not compiled in";', [
            "    } else if (accelerator == GPU_hipFFT) {",
            "#ifdef PSDR_HIP",
            "        fft = std::make_unique<hipFFT>(fft_size, fft_threads, downsample_levels, brightness_offset);",
            "#else",
            '        throw "HIP support is not compiled in";',
            "#endif"], offset=2)),
    ("meson.build", lambda L: after(L, "add_project_arguments('-DCLFFT', language : 'cpp')", [
        "",
        "# MI355X back-end: libpsdr_hip.so + include/psdr.h + phantomsdr_amd/host/*.h (-Dpsdr_dir=<checkout of the HIP core>)",
        "psdr_dir = get_option('psdr_dir')",
        "if psdr_dir != ''",
        "    psdr_dep = declare_dependency(",
        "        dependencies : meson.get_compiler('cpp').find_library('psdr_hip', dirs : psdr_dir / 'phantomsdr_amd'),",
        "        include_directories : include_directories(psdr_dir / 'include', psdr_dir / 'phantomsdr_amd' / 'host'))",
        "    fft_deps += psdr_dep",
        "    add_project_arguments('-DPSDR_HIP', language : 'cpp')",
        "endif"], offset=2)),
]
n1 = {"meson_options.txt": "option('psdr_dir', type : 'string', value : '', description : 'checkout of the MI355X HIP core "
                           "(libpsdr_hip.so built in phantomsdr_amd/)')\n"}


# ------------------------------------------------------------------------------------ level 2
def spectrumserver_h(L):
    L = after(L, '#include "fft.h"', H(["class HipFanout;  // hip_fanout.h: Level 2 of the MI355X back-end",
                                         "namespace psdr_level2 { struct Access; }  // hip_level2.h"]))
    return after(L, "std::unique_ptr<FFT> fft;", H([
        '    std::unique_ptr<HipFanout> fanout;  // all per-client DSP on the GPU (accelerator = "hip", hip_fanout = true)',
        "    void fft_task_hip();                // src/fft_hip.cpp",
        "    friend struct psdr_level2::Access;  // the body of fft_task_hip (hip_level2.h)"]))


def spectrumserver_cpp(L):
    return after(L, "fft->set_output_additional_size(audio_max_fft_size);", H([
        '    if (accelerator == GPU_hipFFT && config["input"]["hip_fanout"].value_or(true)) {',
        "        HipFanout::Params hp{};",
        "        hp.fft_size = (uint32_t)fft_size;",
        "        hp.is_real = is_real;",
        "        hp.downsample_levels = downsample_levels;",
        "        hp.brightness_offset = brightness_offset;",
        "        hp.audio_max_fft_size = audio_max_fft_size;",
        "        hp.audio_max_sps = audio_max_sps;",
        "        hp.skip_num = std::max(1, (int)floor(((float)sps / fft_size) / 10.) * 2);",
        "        hp.min_waterfall_fft = min_waterfall_fft;",
        '        const std::string fmt = config["input"]["driver"]["format"].value_or("f32");',
        '        hp.input_format = fmt == "u8" ? PSDR_FMT_U8 : fmt == "s8" ? PSDR_FMT_S8 : fmt == "u16" ? PSDR_FMT_U16',
        '                        : fmt == "s16" ? PSDR_FMT_S16 : fmt == "f64" ? PSDR_FMT_F64 : PSDR_FMT_F32;',
        '        hp.max_audio_clients = config["limits"]["audio"].value_or(1000);',
        '        hp.max_waterfall_clients = config["limits"]["waterfall"].value_or(1000);',
        '        hp.post_chain = config["input"]["hip_post_chain"].value_or(true);',
        "        hp.ring_halves = 8;",
        "        // more than one GPU of the node: input.hip_devices = [0, 1, ...] (device 0 keeps the ring, the FFT and the waterfall",
        "        // clients, the audio clients are spread over all of them, the spectrum crosses xGMI once per frame: psdr_group_*)",
        '        if (auto *devs = config["input"]["hip_devices"].as_array())',
        "            for (auto &d : *devs) hp.devices.push_back((int)d.value_or<int64_t>(0));",
        "        fanout = std::make_unique<HipFanout>(hp);",
        "    }"]))


def fft_cpp(L):
    return after(L, "void broadcast_server::fft_task() {", H([
        "    if (fanout) {  // Level 2: the whole per-frame path on the GPU (src/fft_hip.cpp)",
        "        fft_task_hip();",
        "        return;",
        "    }"]))


def websocket_cpp(L):
    L = after(L, "client->set_audio_demodulation(default_mode);", H([
        "    if (fanout) client->psdr_attach(fanout.get());  // allocates the GPU-side client slot"]), offset=0)
    return after(L, "client->set_waterfall_range(downsample_levels - 1, 0, min_waterfall_fft);", H([
        "    if (fanout) client->psdr_attach(fanout.get());"]), offset=0)


def signal_h(L):
    L = after(L, "void send_audio(std::complex<float> *buf, size_t frame_num);", H([
        "    void psdr_attach(HipFanout *fo);                       // src/signal.cpp",
        "    void send_audio_hip(HipFanout *fo, size_t frame_num);  // src/fft_hip.cpp",
        "    HipFanout *psdr_fo = nullptr;",
        "    int psdr_id = -1;",
        "    friend struct psdr_level2::Access;                     // the body of send_audio_hip (hip_level2.h)"]))
    return after(L, '#include "client.h"', H(['#include "hip_fanout.h"', "namespace psdr_level2 { struct Access; }"]))


def signal_cpp(L):
    # every state change is forwarded to the GPU-side client
    L = after(L, "void AudioClient::set_audio_range(int l, double m, int r) {", H([
        "    if (psdr_fo) (void)psdr_fo->set_audio_range(psdr_id, l, m, r);  // never throws; a rejected window keeps the old one on the GPU, and the packets carry the labels of the window that was demodulated (hip_level2.h)"]))
    L = after(L, "void AudioClient::set_audio_demodulation(demodulation_mode demodulation) {", H([
        "    if (psdr_fo) (void)psdr_fo->set_audio_demodulation(psdr_id, (psdr_mode)demodulation);"]))
    L = after(L, "    this->agc.reset();", H([
        "    if (psdr_fo) (void)psdr_fo->set_audio_demodulation(psdr_id, (psdr_mode)this->demodulation);  // also resets the GPU AGC"]))
    L = after(L, "void AudioClient::on_close() {", H(["    if (psdr_fo) psdr_fo->remove_audio_client(psdr_id);"]))
    # (the file has no trailing newline: appending would rewrite its last line - insert above the destructor)
    return after(L, "AudioClient::~AudioClient() {", H([
        "void AudioClient::psdr_attach(HipFanout *fo) {",
        "    psdr_fo = fo;",
        "    psdr_id = fo->add_audio_client();  // -1: every GPU slot is taken - this client gets no audio",
        "    (void)fo->set_audio_demodulation(psdr_id, (psdr_mode)demodulation);",
        "}"]), offset=0)


def waterfall_h(L):
    L = after(L, "void send_waterfall(int8_t *buf, size_t frame_num);", H([
        "    void psdr_attach(HipFanout *fo);",
        "    void send_waterfall_hip(HipFanout *fo, size_t frame_num);  // src/fft_hip.cpp",
        "    HipFanout *psdr_fo = nullptr;",
        "    int psdr_id = -1;",
        "    friend struct psdr_level2::Access;                         // the body of send_waterfall_hip (hip_level2.h)"]))
    return after(L, '#include "client.h"', H(['#include "hip_fanout.h"', "namespace psdr_level2 { struct Access; }"]))


def waterfall_cpp(L):
    L = after(L, "    this->level = level;", H([
        "    if (psdr_fo && psdr_id >= 0) psdr_waterfall_set_range(psdr_fo->context(), psdr_id, level, l, r);"]))
    L = after(L, "void WaterfallClient::on_close() {", H(["    if (psdr_fo) psdr_fo->remove_waterfall_client(psdr_id);"]))
    return after(L, "void WaterfallClient::on_close() {", H([
        "void WaterfallClient::psdr_attach(HipFanout *fo) {",
        "    psdr_fo = fo;",
        "    psdr_id = fo->add_waterfall_client();",
        "    if (psdr_id >= 0) psdr_waterfall_set_range(fo->context(), psdr_id, level, l, r);",
        "}"]), offset=0)


def meson2(L):
    return after(L, "cuda_srcs += files('src/fft_cuda.cu')", [
        "",
        "# Level 2 of the MI355X back-end (psdr_dir: see the Level-1 hunk below)",
        "if get_option('psdr_dir') != ''",
        "    cuda_srcs += files('src/fft_hip.cpp')",
        "endif"], offset=2)


e2 = [("src/spectrumserver.h", spectrumserver_h), ("src/spectrumserver.cpp", spectrumserver_cpp), ("src/fft.cpp", fft_cpp),
      ("src/websocket.cpp", websocket_cpp), ("src/signal.h", signal_h), ("src/signal.cpp", signal_cpp),
      ("src/waterfall.h", waterfall_h), ("src/waterfall.cpp", waterfall_cpp), ("meson.build", meson2)]

if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "integration"), exist_ok=True)
    print(build(e1, n1, os.path.join(ROOT, "integration", "level1.patch")).count("\n"), "lines -> integration/level1.patch")
    print(build(e2, {}, os.path.join(ROOT, "integration", "level2.patch")).count("\n"), "lines -> integration/level2.patch")
