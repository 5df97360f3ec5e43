"""Writes tests/golden/plugin_interface.json from the reference's own plugin base module.

Run in the build container only (the reference tree is not on the GPU box):

    python tests/golden/make_plugin_interface.py [/root/reference]

`recommenders/base.py` is loaded BY PATH -- it has no third-party imports, whereas the package's `__init__`
pulls in `implicit` -- and what the serving layer relies on is recorded as data: the abstract methods and every
public method's signature, the result dataclasses' fields and `to_dict` outputs, the `health_check` dictionaries in
its four situations, the exception hierarchy and the messages of the optional hooks.  A CPU test
(tests/test_host.py) holds this package's interface module and plugin class to the file.
"""
import dataclasses
import importlib.util
import inspect
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
SRC = os.path.join(REF, "src", "shrekommender_system", "recommenders", "base.py")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.dont_write_bytecode = True        # nothing is written next to the reference's sources
    spec = importlib.util.spec_from_file_location("reference_recommenders_base", SRC)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod          # dataclasses looks the module up while the class body runs
    spec.loader.exec_module(mod)
    Base = mod.BaseRecommender

    def sig(fn):
        return str(inspect.signature(fn))

    methods = {n: sig(f) for n, f in inspect.getmembers(Base, inspect.isfunction) if not n.startswith("_") or n == "__init__"}
    out = {
        "source": "src/shrekommender_system/recommenders/base.py",
        "abstract_methods": sorted(Base.__abstractmethods__),
        "method_signatures": methods,
        "dataclasses": {},
        "exceptions": {},
    }
    for name in ("RecommendationResult", "SimilarItemsResult"):
        cls = getattr(mod, name)
        out["dataclasses"][name] = [
            {"name": f.name, "type": str(f.type),
             "default": None if f.default is dataclasses.MISSING else repr(f.default),
             "has_default": f.default is not dataclasses.MISSING}
            for f in dataclasses.fields(cls)]
    rec = mod.RecommendationResult("u1", "known", "ALS", [{"movie_id": "m2", "score": 0.5, "rank": 1}], 1, "als@v1")
    sim_ok = mod.SimilarItemsResult("m1", "movie", [{"movie_id": "m3", "similarity_score": 0.25, "rank": 1}], 1, "als@v1")
    sim_err = mod.SimilarItemsResult("mX", "movie", [], 0, "als@v1", error="Movie not found in training data")
    out["to_dict"] = {"recommendation": rec.to_dict(), "similar_ok": sim_ok.to_dict(), "similar_error": sim_err.to_dict()}
    for name in ("RecommenderException", "ModelNotLoadedException", "UserNotFoundException", "ItemNotFoundException"):
        out["exceptions"][name] = [b.__name__ for b in getattr(mod, name).__mro__[1:-1]]

    class Stub(Base):
        stats = {"n_users": 2}

        def load_model(self, model_dir):
            self.is_loaded = True

        def recommend_for_user(self, user_id, N=10):
            return None

        def get_similar_items(self, item_id, N=10):
            return None

        def get_user_profile(self, user_id):
            return {}

        def get_model_stats(self):
            if isinstance(self.stats, Exception):
                raise self.stats
            return self.stats

    s = Stub({"k": 1}, "stub@v1")
    health = {"not_loaded": s.health_check()}
    s.load_model("x")
    health["healthy"] = s.health_check()
    s.stats = {}
    health["empty_stats"] = s.health_check()
    s.stats = RuntimeError("boom")
    health["stats_raise"] = s.health_check()
    out["health_check"] = health
    out["constructor_state"] = {"config": s.config, "model_name": s.model_name, "logger_suffix": s.logger.name.rsplit(".", 1)[-1]}
    hooks = {"warmup": s.warmup(), "teardown": s.teardown()}
    for name, call in (("save_model", lambda: s.save_model("d")), ("train", lambda: s.train({}))):
        try:
            call()
            hooks[name] = "returned"
        except NotImplementedError as exc:
            hooks[name] = {"raises": "NotImplementedError", "message": str(exc)}
    out["optional_hooks"] = hooks
    try:
        Base({}, "x")
        out["instantiating_base"] = "allowed"
    except TypeError:
        out["instantiating_base"] = "TypeError"
    path = os.path.join(HERE, "plugin_interface.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
