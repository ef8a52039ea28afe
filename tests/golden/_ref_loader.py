"""Shared by the fixture generators (build container only): put the REFERENCE's scripts on the import path with the
un-vendored `longcat_video.*` names stubbed, and load the ones whose text is cut off.

Four reference files end in the middle of `main()`'s summary dictionary in this snapshot (`run_delta_a.py`, `run_delta_c.py`,
`run_film_tta.py`, `run_norm_tune_tta.py`: SyntaxError at the last line), so a plain `import` raises.  Everything the fixtures
need — the wrapper classes and the optimise loops — stands ABOVE `def main`; `load_reference_module` executes the file's own
text up to that line, in a module object, from /root/reference.  Nothing of it is written anywhere.
"""
import sys
import types
from pathlib import Path

REF = Path("/root/reference")


def stub_longcat():
    names = ["longcat_video", "longcat_video.modules", "longcat_video.modules.scheduling_flow_match_euler_discrete",
             "longcat_video.modules.autoencoder_kl_wan", "longcat_video.modules.longcat_video_dit",
             "longcat_video.pipeline_longcat_video", "longcat_video.modules.lora_utils",
             "longcat_video.context_parallel", "longcat_video.context_parallel.context_parallel_util"]
    for n in names:
        sys.modules[n] = types.ModuleType(n)
    sys.modules["longcat_video.modules.scheduling_flow_match_euler_discrete"].FlowMatchEulerDiscreteScheduler = object
    sys.modules["longcat_video.modules.autoencoder_kl_wan"].AutoencoderKLWan = object
    sys.modules["longcat_video.modules.longcat_video_dit"].LongCatVideoTransformer3DModel = object
    sys.modules["longcat_video.pipeline_longcat_video"].LongCatVideoPipeline = object
    sys.modules["longcat_video.pipeline_longcat_video"].retrieve_latents = lambda x: x
    sys.modules["longcat_video.modules.lora_utils"].LoRAModule = object


def add_reference_paths():
    for d in ("delta_experiment/scripts", "lora_experiment/scripts"):
        p = str(REF / d)
        if p not in sys.path:
            sys.path.insert(0, p)


def load_reference_module(rel: str, name: str):
    """Import `/root/reference/<rel>`; when its text is cut off inside main(), execute the part above `def main`."""
    path = REF / rel
    src = path.read_text()
    try:
        code = compile(src, str(path), "exec")
        cut = None
    except SyntaxError:
        lines = src.split("\n")
        cut = max(i for i, l in enumerate(lines) if l.startswith("def main("))
        code = compile("\n".join(lines[:cut]), str(path), "exec")
    mod = types.ModuleType(name)
    mod.__file__ = str(path)
    mod.__truncated_at__ = cut
    sys.modules[name] = mod
    exec(code, mod.__dict__)
    return mod
