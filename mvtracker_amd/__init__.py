"""MI355X-native multi-view point-tracking forward path (drop-in for mvtracker.models' predictor).

    from mvtracker_amd import MVTracker, EvaluationPredictor, load_mvtracker

``synth`` (numpy only) can be imported without the HIP library; everything else loads
libmvtracker_hip.so on import and raises if it is missing -- there is no CPU fallback.
"""
__all__ = ["MVTracker", "EvaluationPredictor", "load_mvtracker", "hip", "synth", "sample_io", "adapter", "geometry", "parallel"]


def __getattr__(name):
    if name == "MVTracker":
        from .tracker import MVTracker
        return MVTracker
    if name == "EvaluationPredictor":
        from .predictor import EvaluationPredictor
        return EvaluationPredictor
    if name == "load_mvtracker":
        from .factory import load_mvtracker
        return load_mvtracker
    if name in ("hip", "synth", "sample_io", "adapter", "geometry", "parallel"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
