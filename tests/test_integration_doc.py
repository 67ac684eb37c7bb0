"""The binding INTEGRATION.md documents must compile against include/whisper_amd.h: the C++ stub of section 1 is cut out of
the document and type-checked by g++ with a stand-in for the one reference header it includes (System.hpp: the class shell
and the members the stub names).  A stale argument list in the document fails here."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SYSTEM_HPP = r"""
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <iostream>
#include "whisper_amd.h"
template <typename URV> class System {
public:
  bool gpuBatchRun(const std::vector<std::string>& elfs, const std::string& isa, uint64_t memBase, uint64_t memSize,
                   uint64_t instLimit, std::vector<wsp_result>& results);
  unsigned hartIdOffset_ = 0;
};
static unsigned tlbEntries = 0, bytesPerVec = 0;
"""


def _cpp_blocks(text):
    return re.findall(r"```cpp\n(.*?)```", text, re.S)


def test_documented_cpp_binding_compiles_against_the_header():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = _cpp_blocks(doc)
    assert blocks, "INTEGRATION.md holds no C++ stub"
    stub = blocks[0]
    called = set(re.findall(r"\b(wsp_[a-z_]+)\s*\(", stub))
    assert {"wsp_batch_create", "wsp_batch_load_elf", "wsp_batch_set_stop", "wsp_batch_run", "wsp_batch_results",
            "wsp_batch_destroy", "wsp_last_error"} <= called
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "System.hpp"), "w").write(SYSTEM_HPP)
        src = os.path.join(td, "GpuBatch.cpp")
        open(src, "w").write(stub + "\ntemplate class System<uint64_t>;\ntemplate class System<uint32_t>;\n")
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused-variable",
                            "-I", os.path.join(ROOT, "include"), "-I", td, src], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_every_entry_point_the_document_names_is_declared():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    hdr = open(os.path.join(ROOT, "include", "whisper_amd.h")).read()
    declared = set(re.findall(r"\b(wsp_[a-z0-9_]+)\s*\(", hdr))
    for name in set(re.findall(r"\b(wsp_[a-z0-9_]+)\s*\(", doc)):
        assert name in declared, f"INTEGRATION.md names {name}, which include/whisper_amd.h does not declare"
