// Shared by the C-ABI translation units: exception -> error-code boundary.
// The reference throws fl_exception and lets it escape main (flexception.h:8-24); nothing may escape a C ABI,
// so every entry point runs inside guard().
#pragma once
#include "host_scene.h"
#include <new>
#include <string>

namespace lj {

void set_last_error(const std::string &msg);

template <typename F> int guard(F &&f) {
    try {
        f();
        return LJ_OK;
    } catch (const LjError &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc &) {
        set_last_error("out of host memory");
        return LJ_ERR_INTERNAL;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return LJ_ERR_INTERNAL;
    } catch (...) {
        set_last_error("unknown error");
        return LJ_ERR_INTERNAL;
    }
}

} // namespace lj
