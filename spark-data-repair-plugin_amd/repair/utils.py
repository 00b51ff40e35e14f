"""Host-side helpers mirroring python/repair/utils.py of the reference (logger, option parsing,
argument type checks, wall-clock decorator).  Same names, argument meaning and error behaviour;
written for a Spark-free (pandas) host."""
import functools
import inspect
import logging
import os
import time
import typing
from typing import Any, Dict, List, Optional


def setup_logger() -> logging.Logger:
    logger = logging.getLogger("repair")
    if not logger.handlers:
        logger.addHandler(logging.NullHandler())
    return logger


_logger = setup_logger()


def is_testing() -> bool:
    # the reference keys this on SPARK_TESTING (utils.py:229-230); REPAIR_TESTING is an alias
    return os.environ.get("SPARK_TESTING") is not None or os.environ.get("REPAIR_TESTING") is not None


def to_list_str(d: List[Any], sep: str = ",", quote: bool = False) -> str:
    return sep.join(("'%s'" % e) if quote else str(e) for e in d)


def get_option_value(opts: Dict[str, str], key: str, default_value: Any, type_class: Any = str,
                     validator: Optional[Any] = None, err_msg: Optional[str] = None) -> Any:
    """reference utils.py:50-75: cast + validate; bad values raise only under testing, else warn+default."""
    assert type(default_value) is type_class, "key=%s" % key
    if key not in opts:
        return default_value
    raw = opts[key]
    try:
        if type_class is bool and isinstance(raw, str):
            value = raw.strip().lower() not in ("", "0", "false", "no")
        else:
            value = type_class(raw)
    except Exception:
        msg = 'Failed to cast "%s" into %s data: key=%s' % (raw, type_class.__name__, key)
        if is_testing():
            raise ValueError(msg)
        _logger.warning(msg)
        return default_value
    if validator is not None and not validator(value):
        msg = "%s, got %s" % (str(err_msg).format(key), value)
        if is_testing():
            raise ValueError(msg)
        _logger.warning(msg)
        return default_value
    return value


def _type_name(t: Any) -> str:
    origin = getattr(t, "__origin__", None)
    if origin is list:
        return "list[%s]" % _type_name(t.__args__[0])
    if origin is dict:
        return "dict[%s,%s]" % (_type_name(t.__args__[0]), _type_name(t.__args__[1]))
    return getattr(t, "__name__", str(t))


def _matches(v: Any, t: Any) -> bool:
    origin = getattr(t, "__origin__", None)
    if origin is list:
        return type(v) is list and all(_matches(e, t.__args__[0]) for e in v)
    if origin is dict:
        return type(v) is dict and all(_matches(k, t.__args__[0]) for k in v) and all(_matches(x, t.__args__[1]) for x in v.values())
    if origin is typing.Union:
        return any(_matches(v, a) for a in t.__args__)
    if t is Any:
        return True
    return isinstance(v, t)


def argtype_check(f):  # type: ignore
    """Checks call arguments against the function's annotations (reference utils.py:149-216);
    same TypeError messages: "`x` should be provided as <type>, got <type>"."""
    sig = inspect.signature(f)

    @functools.wraps(f)
    def wrapper(self, *args, **kwargs):  # type: ignore
        for name, v in sig.bind(self, *args, **kwargs).arguments.items():
            annot = sig.parameters[name].annotation
            if annot is inspect.Parameter.empty or isinstance(annot, str):
                continue
            origin = getattr(annot, "__origin__", None)
            if origin is typing.Union:
                if not _matches(v, annot):
                    want = "/".join(_type_name(a) for a in annot.__args__)
                    raise TypeError("`%s` should be provided as %s, got %s" % (name, want, type(v).__name__))
            elif origin is list:
                if type(v) is not list:
                    raise TypeError("`%s` should be provided as %s, got %s" % (name, _type_name(annot), type(v).__name__))
                bad = [e for e in v if not _matches(e, annot.__args__[0])]
                if bad:
                    raise TypeError("`%s` should be provided as %s, got %s in elements" % (name, _type_name(annot), type(bad[0]).__name__))
            elif origin is dict:
                if type(v) is not dict:
                    raise TypeError("`%s` should be provided as %s, got %s" % (name, _type_name(annot), type(v).__name__))
                badk = [k for k in v if not _matches(k, annot.__args__[0])]
                if badk:
                    raise TypeError("`%s` should be provided as %s, got %s in keys" % (name, _type_name(annot), type(badk[0]).__name__))
                badv = [x for x in v.values() if not _matches(x, annot.__args__[1])]
                if badv:
                    raise TypeError("`%s` should be provided as %s, got %s in values" % (name, _type_name(annot), type(badv[0]).__name__))
            elif not _matches(v, annot):
                raise TypeError("`%s` should be provided as %s, got %s" % (name, _type_name(annot), type(v).__name__))
        return f(self, *args, **kwargs)

    return wrapper


def elapsed_time(f):  # type: ignore
    """reference utils.py:219-226: returns (result, seconds)."""
    @functools.wraps(f)
    def wrapper(*args, **kwargs):  # type: ignore
        t0 = time.time()
        ret = f(*args, **kwargs)
        return ret, time.time() - t0

    return wrapper


def job_group(name: str):  # type: ignore
    """Stand-in for the reference's spark_job_group (utils.py:130-146): logs the phase's elapsed time."""
    def deco(f):  # type: ignore
        @functools.wraps(f)
        def wrapper(*args, **kwargs):  # type: ignore
            t0 = time.time()
            ret = f(*args, **kwargs)
            _logger.info("Elapsed time (name: %s) is %s(s)" % (name, time.time() - t0))
            return ret
        return wrapper
    return deco


# ---------------------------------------------------------------------------------------------------------------------
# One hash pass per object column and run.  NULL detection (`isna`), the domain statistics (`nunique`) and the dictionary encoding
# of the resident path (`pipeline.encode_frame`) each walk the 10^6 Python objects of a string column; inside `column_code_cache`
# the first of them factorises the column and the others read the codes (NULL = -1) and the dictionary.
# ---------------------------------------------------------------------------------------------------------------------
import threading as _threading

_frame_codes = _threading.local()


class column_code_cache:
    """Context manager: while active (per thread), the column_* helpers below share one `pandas.factorize` per object column of
    `df`.  The frame must not be modified inside the block (RepairModel.run never writes into its input)."""

    def __init__(self, df: Any) -> None:
        self.df = df

    def __enter__(self) -> "column_code_cache":
        self.prev = getattr(_frame_codes, "cur", None)
        _frame_codes.cur = (self.df, {})
        return self

    def __exit__(self, *a: Any) -> None:
        _frame_codes.cur = self.prev


def _code_store(df: Any) -> Optional[Dict[str, Any]]:
    cur = getattr(_frame_codes, "cur", None)
    return cur[1] if cur is not None and cur[0] is df else None


def column_factorize(df: Any, col: str) -> Any:
    """(codes int32 with -1 = NULL, distinct values in order of first appearance) of a NON-numeric column."""
    import numpy as np
    import pandas as pd
    store = _code_store(df)
    if store is not None and col in store:
        return store[col]
    s = df[col]
    if isinstance(s.dtype, pd.api.extensions.ExtensionDtype):
        # categorical / Arrow-backed / pandas string columns carry their dictionary or a contiguous buffer already: factorize works on
        # that (3-15 ms per 10^6 cells) instead of on 10^6 Python objects
        codes, uniq = pd.factorize(s, use_na_sentinel=True)
    else:
        codes, uniq = pd.factorize(s.to_numpy(dtype=object), use_na_sentinel=True)
    out = (codes.astype(np.int32, copy=False), np.asarray(uniq, dtype=object))
    if store is not None:
        store[col] = out
    return out


def _shares_codes(df: Any, col: str) -> bool:
    return _code_store(df) is not None and df[col].dtype == object


def column_isna(df: Any, col: str) -> Any:
    """Boolean numpy mask of the NULL cells of a column."""
    if _shares_codes(df, col):
        return column_factorize(df, col)[0] < 0
    return df[col].isna().to_numpy()


def column_nunique(df: Any, col: str) -> int:
    """Number of distinct non-NULL values of a column."""
    if _shares_codes(df, col):
        return int(len(column_factorize(df, col)[1]))
    return int(df[col].nunique(dropna=True))
