"""Weights-only reader for the reference's checkpoints (format: yolo/engine/trainer.py:411-436; loader it replaces: nn/tasks.py:520-601).

The reference pickles whole nn.Module objects (`ckpt['model']`, `ckpt['ema']`, stored in fp16), so `torch.load` must import and run
`ultralytics.*` classes - and `torch.load(weights_only=True)` refuses the file.  This reader walks the same pickle stream with an
Unpickler whose `find_class` resolves NOTHING from the file except plain containers and tensor rebuilding: every other global (the model
classes, `copyreg._reconstructor`, argparse namespaces, ...) becomes an inert stub object that only records the state it is given.  No
code named by the file is imported or called.  From the stub tree it recovers what `attempt_load_one_weight` needs: the model's YAML dict,
class names, `args`, and the state_dict (module tree walk over `_parameters` / `_buffers` / `_modules`)."""
import io
import pickle
import zipfile
from collections import OrderedDict

import torch

_DTYPES = {'FloatStorage': torch.float32, 'HalfStorage': torch.float16, 'BFloat16Storage': torch.bfloat16, 'DoubleStorage': torch.float64,
           'LongStorage': torch.int64, 'IntStorage': torch.int32, 'ShortStorage': torch.int16, 'CharStorage': torch.int8, 'ByteStorage': torch.uint8,
           'BoolStorage': torch.bool}
_SAFE_BUILTINS = {'set': set, 'frozenset': frozenset, 'list': list, 'dict': dict, 'tuple': tuple, 'int': int, 'float': float, 'bool': bool, 'str': str,
                  'bytes': bytes, 'complex': complex, 'slice': slice, 'range': range, 'bytearray': bytearray}


class Stub:
    """Stands in for any object whose class the file names: remembers constructor arguments and state, runs nothing."""

    def __init__(self, *args, **kwargs):
        self.__dict__['_args'] = args

    def __setstate__(self, state):
        self.__dict__['_state'] = state

    def __setitem__(self, k, v):                 # OrderedDict-like subclasses are filled with SETITEMS
        self.__dict__.setdefault('_items', OrderedDict())[k] = v

    def append(self, v):
        self.__dict__.setdefault('_list', []).append(v)

    def extend(self, vs):
        self.__dict__.setdefault('_list', []).extend(vs)

    @property
    def state(self):
        st = self.__dict__.get('_state')
        return st if isinstance(st, dict) else {}


class _StorageType:
    def __init__(self, name):
        self.dtype = _DTYPES[name]


def _reconstructor(cls, base, state):            # copyreg._reconstructor(cls, object, None) of the module objects
    return cls() if isinstance(cls, type) else Stub()


def _load_type(name):
    """dill's spelling of a builtin type (`dill._dill._load_type('OrderedDict')`; the reference pickles with dill when it is importable,
    trainer.py:424-428): the same allow-list as a GLOBAL opcode, anything else an inert stub class."""
    if name == 'OrderedDict':
        return OrderedDict
    if name in _SAFE_BUILTINS:
        return _SAFE_BUILTINS[name]
    return type(str(name), (Stub,), {'__module__': 'stub:dill'})


def _inert(*args, **kwargs):                     # dill._dill._create_function / _create_code / _create_cell ...: lambdas in the file become stubs
    return Stub(*args)


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    flat = storage                                 # 1-D tensor over the whole storage (see persistent_load)
    if len(size) == 0:
        return flat[storage_offset].clone()
    return torch.as_strided(flat, tuple(size), tuple(stride), storage_offset).clone()


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None, *rest):
    return data


class _Reader(pickle.Unpickler):
    def __init__(self, f, zf, prefix):
        super().__init__(f)
        self.zf, self.prefix, self.stubbed = zf, prefix, set()

    def find_class(self, module, name):
        if module == 'collections' and name == 'OrderedDict':
            return OrderedDict
        if module in ('builtins', '__builtin__') and name in _SAFE_BUILTINS:
            return _SAFE_BUILTINS[name]
        if module == 'torch._utils' and name == '_rebuild_tensor_v2':
            return _rebuild_tensor_v2
        if module == 'torch._utils' and name in ('_rebuild_parameter', '_rebuild_parameter_with_state'):
            return _rebuild_parameter
        if module == 'torch' and name in _DTYPES:
            return _StorageType(name)
        if module == 'torch' and name == 'Size':
            return tuple
        if module == 'torch' and name in ('float16', 'float32', 'float64', 'bfloat16', 'int64', 'int32', 'uint8', 'bool'):
            return getattr(torch, name)
        if module == 'copyreg' and name == '_reconstructor':
            return _reconstructor
        if module in ('dill._dill', 'dill.dill'):
            self.stubbed.add(f'{module}.{name}')
            return _load_type if name == '_load_type' else _inert
        self.stubbed.add(f'{module}.{name}')
        return type(name, (Stub,), {'__module__': f'stub:{module}'})       # a fresh inert class; nothing is imported

    def persistent_load(self, pid):
        # ('storage', storage_type, key, location, numel) of torch.save's zip format
        if not (isinstance(pid, tuple) and pid and pid[0] == 'storage'):
            raise pickle.UnpicklingError(f'unexpected persistent id {pid!r}')
        st, key, numel = pid[1], pid[2], pid[4]
        dtype = st.dtype if isinstance(st, _StorageType) else getattr(st, 'dtype', torch.uint8)
        raw = self.zf.read(f'{self.prefix}/data/{key}' if self.prefix else f'data/{key}')
        if numel == 0 or len(raw) == 0:
            return torch.zeros(0, dtype=dtype)
        return torch.frombuffer(bytearray(raw), dtype=dtype)


def read_checkpoint(path):
    """-> (object tree with Stubs, sorted list of the globals that were stubbed out)."""
    if not zipfile.is_zipfile(path):
        raise RuntimeError(f'{path}: not a torch.save zip archive (the legacy tar/pickle formats are not read)')
    with zipfile.ZipFile(path) as zf:
        pkl = next((n for n in zf.namelist() if n.endswith('/data.pkl') or n == 'data.pkl'), None)
        if pkl is None:
            raise RuntimeError(f'{path}: no data.pkl inside')
        prefix = pkl[:-len('/data.pkl')] if '/' in pkl else ''
        rd = _Reader(io.BytesIO(zf.read(pkl)), zf, prefix)
        return rd.load(), sorted(rd.stubbed)


def module_state_dict(mod, prefix=''):
    """state_dict of a stubbed nn.Module tree: persistent buffers + parameters, in module order (nn.Module.state_dict semantics)."""
    out = OrderedDict()
    st = mod.state if isinstance(mod, Stub) else {}
    skip = st.get('_non_persistent_buffers_set') or set()
    for name, p in (st.get('_parameters') or {}).items():
        if torch.is_tensor(p):
            out[prefix + name] = p
    for name, b in (st.get('_buffers') or {}).items():
        if torch.is_tensor(b) and name not in skip:
            out[prefix + name] = b
    for name, child in (st.get('_modules') or {}).items():
        if child is not None:
            out.update(module_state_dict(child, f'{prefix}{name}.'))
    return out


def plain(obj):
    """Stub namespaces (e.g. the pickled train args) -> plain dicts; containers recursively."""
    if isinstance(obj, Stub):
        return {k: plain(v) for k, v in obj.state.items()}
    if isinstance(obj, dict):
        return {k: plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(plain(v) for v in obj)
    return obj
