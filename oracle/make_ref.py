"""TEST / BASELINE INFRASTRUCTURE ONLY - build recipe of ``oracle/_ref/``: the reference's hot-path modules, COMPILED.

    python oracle/make_ref.py            (run by __graft_entry__.build() wherever /root/reference exists)

``/root/reference`` does not exist on the GPU box, and reference SOURCES are never copied into this repository.  What a C
reference gets - compiled from its own sources where they lie, outputs only into ``oracle/_ref/`` (git-ignored, shipped by
gpurun like our own built ``.so``) - is done here for the Python reference: ``model.py``, ``embedding.py``, ``render_rays.py``
and ``loss.py`` (the four files of the path that need nothing but torch + numpy) are byte-compiled
UNMODIFIED from ``/root/reference`` into sourceless ``oracle/_ref/<name>.pyc`` modules, next to a ``MANIFEST.json`` holding
the sha256 of each source file, the interpreter's bytecode magic and the torch version of the build.  ``oracle/ref_runner.py``
imports them from there when the source tree is absent; so the GPU box runs THE REFERENCE'S OWN code (functorch ``vmap`` +
``torch.optim.AdamW`` driven the way utils.py:30-34 / train.py:293-326 drive it) as

  * ``bench.py``'s ``cpu_baseline`` (``"kind": "reference"``) and ``gpu_reference_baseline`` (the north star's denominator),
  * the ``-m gpu`` test that regenerates the ``tiny`` fixture on the box and compares it with the committed one.

Never imported by the product package.
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
MODULES = ("model", "embedding", "render_rays", "loss")
MANIFEST = os.path.join(REF_DIR, "MANIFEST.json")


def _sha256(path):
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def build(reference_root="/root/reference", force=False):
    """Compile the four modules; returns the manifest (dict) or None when the reference tree is absent (the GPU box: the prebuilt
    files travel with the snapshot)."""
    if not os.path.isdir(reference_root):
        return load_manifest()
    srcs = {m: os.path.join(reference_root, m + ".py") for m in MODULES}
    shas = {m: _sha256(p) for m, p in srcs.items()}
    old = load_manifest()
    magic = importlib.util.MAGIC_NUMBER.hex()
    if (not force and old and old.get("sha256") == shas and old.get("bytecode_magic") == magic
            and all(os.path.exists(os.path.join(REF_DIR, m + ".pyc")) for m in MODULES)):
        return old
    os.makedirs(REF_DIR, exist_ok=True)
    for m, p in srcs.items():
        # dfile: the path recorded in tracebacks / co_filename - the ORIGINAL location, so a traceback on the GPU box still cites
        # /root/reference/<file>:<line>; UNCHECKED_HASH: the .pyc is valid without its source file next to it
        py_compile.compile(p, cfile=os.path.join(REF_DIR, m + ".pyc"), dfile=p, doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    import torch
    man = {"what": "kxhit/vMAP hot-path modules byte-compiled unmodified from reference_root (no source is copied)",
           "basis": "a build artefact of the reference's own files, produced where they lie by this committed recipe and written only under "
                    "oracle/_ref/ (git-ignored; never committed, never imported by the product package vmap_amd/) - the Python counterpart "
                    "of compiling a C reference into oracle/_ref/*.so; the reference's licence file (if any) stays with its sources at "
                    "reference_root, and the modules are used only as the checker / timed baseline",
           "reference_root": reference_root, "modules": list(MODULES), "sha256": shas, "bytecode_magic": magic,
           "python": sys.version.split()[0], "torch_version_at_build": torch.__version__}
    with open(MANIFEST, "w") as fh:
        json.dump(man, fh, indent=1, sort_keys=True)
    return man


def load_manifest():
    try:
        with open(MANIFEST) as fh:
            return json.load(fh)
    except Exception:
        return None


def why_unavailable():
    """None when the compiled modules are there AND this interpreter can load them; else the reason, in words (bench.py prints it
    instead of a bare null when the north star's denominator cannot be measured on a box)."""
    man = load_manifest()
    if not man:
        return f"no {os.path.relpath(MANIFEST, os.path.dirname(HERE))} (oracle/make_ref.py builds oracle/_ref where /root/reference exists)"
    if man.get("bytecode_magic") != importlib.util.MAGIC_NUMBER.hex():
        return (f"bytecode magic mismatch: oracle/_ref was compiled by Python {man.get('python')} (magic {man.get('bytecode_magic')}), "
                f"this interpreter is {sys.version.split()[0]} (magic {importlib.util.MAGIC_NUMBER.hex()})")
    missing = [m for m in MODULES if not os.path.exists(os.path.join(REF_DIR, m + ".pyc"))]
    if missing:
        return f"oracle/_ref lacks {missing}"
    return None


def available():
    """True when the compiled modules are there AND this interpreter can load them (same bytecode magic)."""
    return why_unavailable() is None


if __name__ == "__main__":
    print(json.dumps(build(force="--force" in sys.argv), indent=1))
