"""RISCOF DUT plugin for whisper_amd: the role of /root/reference/arch_test_target/riscof_whisper.py, with one difference that is the
point of this library - the reference plugin starts one `whisper --isa=.. --testsignature=.. --target=..` process per test
(riscof_whisper.py:152), this one compiles the tests the same way and then runs *all* of them in one batch, one hart per test,
on the GPU, and writes every DUT-<name>.signature file in the reference's format (System::produceTestSignatureFile).

config.ini:
    [whisper_amd]
    pluginpath = <this directory>
    ispec = <.../whisper_isa.yaml>        # the reference's arch_test_target yaml files work unchanged
    pspec = <.../whisper_platform.yaml>
    env = <directory with link.ld / model_test.h>   (default: pluginpath)

riscof itself is not part of this image: the class below is only defined when riscof can be imported; `run_batch` (what runTests
calls after compiling) has no riscof dependency and is what the tests exercise."""
from __future__ import annotations

import os

from whisper_amd import cli


def isa_string(ispec_isa: str, xlen: str) -> str:
    """riscv-config ISA string -> whisper --isa string, as riscof_whisper.py:93-99 builds it - except that the letters are looked
    for behind the "RV64" / "RV32" prefix: the reference's `"V" in ispec["ISA"]` also matches the V of "RV64" and so always asks
    for the vector extension."""
    isa = "rv" + xlen
    ispec_isa = ispec_isa[4:] if ispec_isa[:2].upper() == "RV" else ispec_isa
    for ext in ["I", "M", "A", "F", "D", "C", "V", "S", "U"]:
        if ext in ispec_isa:
            isa += ext.lower()
    for ext in ["Zfa", "Zfh", "Zba", "Zbb", "Zbc", "Zbkb", "Zbkx", "Zbs", "Zknd", "Zkne", "Zknh", "Zksed", "Zksh", "Zicsr"]:
        if ext in ispec_isa:
            isa += "_" + ext.lower()
    return isa


def run_batch(tests, isa, maxinst=None, config=None):
    """tests: [(elf path, signature path)].  One batch, one hart per test.  Returns [ok per test]."""
    return cli.run_entries([(e, s, None) for e, s in tests], isa=isa, config=config, maxinst=maxinst, quiet=True)


try:
    import riscof.utils as utils
    from riscof.pluginTemplate import pluginTemplate
except ImportError:                                   # riscof is absent in this image
    pluginTemplate = None

if pluginTemplate is not None:
    class whisper_amd(pluginTemplate):
        __model__ = "whisper_amd"
        __version__ = "0.2"

        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            config = kwargs.get("config")
            if config is None:
                raise SystemExit("Please enter input file paths in configuration.")
            self.num_jobs = str(config["jobs"] if "jobs" in config else 1)
            self.pluginpath = os.path.abspath(config["pluginpath"])
            self.env = os.path.abspath(config["env"]) if "env" in config else self.pluginpath
            self.isa_spec = os.path.abspath(config["ispec"])
            self.platform_spec = os.path.abspath(config["pspec"])
            self.target_run = not ("target_run" in config and config["target_run"] == "0")

        def initialise(self, suite, work_dir, archtest_env):
            self.work_dir = work_dir
            self.suite_dir = suite
            self.compile_cmd = ("riscv64-unknown-elf-gcc -march={0} -static -mcmodel=medany -fvisibility=hidden -nostdlib "
                                "-nostartfiles -g -T " + self.env + "/link.ld -I " + self.env + "/ -I " + archtest_env + " {2} -o {3} {4}")

        def build(self, isa_yaml, platform_yaml):
            ispec = utils.load_yaml(isa_yaml)["hart0"]
            self.xlen = "64" if 64 in ispec["supported_xlen"] else "32"
            self.isa = isa_string(ispec["ISA"], self.xlen)
            self.compile_cmd += " -mabi=" + ("lp64 " if 64 in ispec["supported_xlen"] else "ilp32 ")

        def runTests(self, testList):
            make = utils.makeUtil(makefilePath=os.path.join(self.work_dir, "Makefile." + self.name[:-1]))
            make.makeCommand = "make -k -j" + self.num_jobs
            tests = []
            for testname in testList:
                t = testList[testname]
                elf = os.path.join(t["work_dir"], "my.elf")
                sig = os.path.join(t["work_dir"], self.name[:-1] + ".signature")
                macros = " -D" + " -D".join(t["macros"])
                make.add_target("@cd {0}; {1};".format(t["work_dir"], self.compile_cmd.format(t["isa"].lower(), self.xlen, t["test_path"], elf, macros)))
                tests.append((elf, sig))
            make.execute_all(self.work_dir, timeout=900)          # compile only: the runs follow, all at once
            if not self.target_run:
                raise SystemExit(0)
            run_batch([t for t in tests if os.path.exists(t[0])], self.isa)
