"""MI355X-native any-to-many generation path for Spider. Product classes are resolved lazily so that
`import spider_amd` stays free of torch / HIP initialisation."""

_LAZY = {"SpiderFreeInfer": "spider_free", "SpiderFreeResult": "spider_free", "SpiderDecoder": "spider_decoder",
         "SpiderDecoderInfer": "spider_decoder", "SpiderStoryFreeInfer": "spider_decoder"}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module("." + _LAZY[name], __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
