"""Model registry with the two methods the reference's callers use (spider/common/registry.py:81-108,234):
`@registry.register_model(name)` and `registry.get_model_class(name)`; classes are then built as
`model_cls(**cfg.model)` after popping `type` (spider_decoder_infer.py:37-41)."""


class Registry:
    mapping = {"model_name_mapping": {}}

    @classmethod
    def register_model(cls, name):
        def wrap(model_cls):
            if name in cls.mapping["model_name_mapping"]:
                raise KeyError(f"Name '{name}' already registered for {cls.mapping['model_name_mapping'][name]}.")
            cls.mapping["model_name_mapping"][name] = model_cls
            return model_cls
        return wrap

    @classmethod
    def get_model_class(cls, name):
        return cls.mapping["model_name_mapping"].get(name, None)

    @classmethod
    def list_models(cls):
        return sorted(cls.mapping["model_name_mapping"].keys())


registry = Registry()
