"""The host half of the C ABI refuses what it must BEFORE anything reaches a device (no GPU needed): every entry point that takes an
espm_mu_state checks the struct's size / ABI version first, pointer and size arguments are validated with a message in
espm_mu_last_error(), and a refused call leaves no state behind.  The same calls run under AddressSanitizer / UBSan against a host-
instrumented build of the library (tools/sanitize/run_host_asan.sh)."""
import ctypes as C

import pytest


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from espm_amd import _lib
    return _lib


def _zero_args(argtypes):
    out = []
    for t in argtypes:
        if t is C.c_void_p or t is C.c_char_p or (isinstance(t, type) and issubclass(t, C._Pointer)):
            out.append(None)
        elif t in (C.c_double, C.c_float):
            out.append(t(0.0))
        else:
            out.append(t(0))
    return out


def _state_functions(lib):
    return sorted(name for name, (res, args) in lib.SYMBOLS.items() if args and args[0] is lib._SP)


def test_every_state_entry_point_refuses_a_foreign_struct(lib):
    """struct_size / abi_version are read before any other field: a zeroed struct, a struct of another size and one of another
    version are refused with ESPM_EINVAL by every entry point that takes a state - also the ones that would launch kernels."""
    names = _state_functions(lib)
    assert len(names) >= 30
    for handle in (lib.lib, lib.variant(12).lib):
        for name in names:
            res, argtypes = lib.SYMBOLS[name]
            fn = getattr(handle, name)
            for tamper in ("zero", "size", "version"):
                st = lib.MUState()
                if tamper == "zero":
                    C.memset(C.byref(st), 0, C.sizeof(st))
                elif tamper == "size":
                    st.struct_size += 8
                else:
                    st.abi_version += 1
                rc = fn(C.byref(st), *_zero_args(argtypes[1:]))
                if res is C.c_size_t:                       # (espm_mu_shard_record_bytes: a size, 0 when refused)
                    assert rc == 0, (name, tamper)
                elif name in ("espm_mu_fused_applies", "espm_mu_w_update_is_local"):   # predicates: "no" for a struct they cannot read
                    assert rc == 0, (name, tamper)
                else:
                    assert rc == lib.EINVAL, (name, tamper, rc)
                    msg = handle.espm_mu_last_error()
                    assert b"struct_size" in msg or b"abi" in msg.lower(), (name, msg)


def test_a_state_that_was_never_queried_is_refused(lib):
    """A struct of the right size and version but without the derived fields (n_pad, p_pad ...: espm_mu_query) or without its buffers
    is refused by the compute entry points with a message that names what is missing - nothing is launched on NULL pointers."""
    for name in ("espm_mu_step_h", "espm_mu_step_hw", "espm_mu_w_accum", "espm_mu_iterate", "espm_mu_build_gw", "espm_mu_hstat", "espm_mu_w_reduce"):
        st = lib.MUState()
        st.n, st.p, st.k, st.x_dtype = 64, 64, 3, lib.X_F32
        res, argtypes = lib.SYMBOLS[name]
        rc = getattr(lib.lib, name)(C.byref(st), *_zero_args(argtypes[1:]))
        assert rc == lib.EINVAL and b"espm_mu_query" in lib.lib.espm_mu_last_error(), (name, rc, lib.lib.espm_mu_last_error())
        assert lib.lib.espm_mu_query(C.byref(st)) == 0
        rc = getattr(lib.lib, name)(C.byref(st), *_zero_args(argtypes[1:]))
        assert rc != 0 and lib.lib.espm_mu_last_error(), (name, rc)      # (no buffers: refused, or the runtime's own error without a device)


def test_exchange_and_helper_entry_points_validate_their_arguments(lib):
    L = lib.lib
    ctx = C.c_void_p()
    for world, rank, nbytes in ((0, 0, 64), (17, 0, 64), (2, 2, 64), (2, -1, 64), (2, 0, 0), (2, 0, 24)):
        assert L.espm_xchg_create(world, rank, nbytes, C.byref(ctx)) == lib.EINVAL and not ctx.value, (world, rank, nbytes)
    assert L.espm_xchg_create(2, 0, 64, None) == lib.EINVAL
    assert L.espm_xchg_handle(None, None) == lib.EINVAL and L.espm_xchg_connect(None, None) == lib.EINVAL
    assert L.espm_xchg_post(None, 1, None) == lib.EINVAL and L.espm_xchg_wait(None, 1, None) == lib.EINVAL
    assert L.espm_xchg_timeouts(None, None) == lib.EINVAL and L.espm_xchg_set_order(None, 1) == lib.EINVAL
    assert L.espm_xchg_order(None) == -1 and L.espm_xchg_staging(None) is None and L.espm_xchg_records(None, 0) is None
    assert L.espm_xchg_destroy(None) == 0
    assert L.espm_lu_pl(None, 0, 8, 4, 4, None, None, 0, None) == lib.EINVAL
    assert L.espm_mu_laplacian(None, 4, 4, 3, 16, None, None) == lib.EINVAL
    assert L.espm_mu_pack_x(None, 0, 0, 0, 4, 4, None, None, 0, 8, 512, 256, 8, None) == lib.EINVAL
    assert L.espm_dichotomy_simplex(None, None, 3, 4, 4, 0.5, 1e-5, 100, None, None, None) != 0
    assert L.espm_mu_last_error()


def test_last_error_is_per_thread(lib):
    """espm_mu_last_error() belongs to the calling thread: a refusal on another thread does not overwrite this one's message."""
    import threading
    st = lib.MUState()
    st.n = 0
    assert lib.lib.espm_mu_query(C.byref(st)) == lib.EINVAL
    mine = lib.lib.espm_mu_last_error()
    seen = []

    def other():
        lib.lib.espm_xchg_create(0, 0, 64, None)
        seen.append(lib.lib.espm_mu_last_error())
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen and seen[0] != mine and lib.lib.espm_mu_last_error() == mine
