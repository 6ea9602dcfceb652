"""`pip install -e .` / `python setup.py build_ext --inplace`: builds lc_amd/_C/liblc_amd.so with hipcc for gfx950 (the same
command `python __graft_entry__.py build` runs) and ships it as package data.  No torch extension machinery: the library
is a plain C-ABI shared object loaded with ctypes (include/lc_amd.h)."""
import os
import sys

from setuptools import Command, find_packages, setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


class build_ext(Command):
    description = "compile lc_amd/csrc/*.hip into lc_amd/_C/liblc_amd.so (hipcc --offload-arch=gfx950)"
    user_options = [("inplace", "i", "kept for the usual spelling; the library is always built in tree"), ("force", "f", "rebuild")]

    def initialize_options(self):
        self.inplace, self.force = 1, 0

    def finalize_options(self):
        pass

    def run(self):
        from lc_amd import build as lc_build

        print("built", lc_build.build(force=bool(self.force), verbose=True))


class build_py_with_lib(build_py):
    def run(self):
        self.run_command("build_ext")
        super().run()


setup(
    name="lc_amd",
    version="0.1.0",
    description="MI355X-native hot path of fulliu/lc: linear-covariance pose loss, weighted PnP, keypoint head (HIP, gfx950)",
    packages=find_packages(include=["lc_amd", "lc_amd.*"]),
    package_data={"lc_amd": ["_C/*.so", "csrc/*.hip", "csrc/*.h"]},
    data_files=[("include", ["include/lc_amd.h"])],
    python_requires=">=3.10",
    install_requires=["torch", "numpy"],
    cmdclass={"build_ext": build_ext, "build_py": build_py_with_lib},
)
