// oracle/_stub/vbem — TEST INFRASTRUCTURE (see parallel_for.h): oneapi::tbb::combinable with ONE thread's copy — the serial schedule.  As in oneTBB the
// constructor's arguments are what each thread's copy is built from, on first use.
#pragma once
#include <functional>
#include <memory>
namespace oneapi { namespace tbb { template <class T> class combinable { std::function<T*()> make_; std::unique_ptr<T> v_; public:
  combinable() : make_([] { return new T(); }) {} template <class... A> explicit combinable(A... a) : make_([=] { return new T(a...); }) {}
  T& local() { if (!v_) v_.reset(make_()); return *v_; } template <class F> void combine_each(F f) { f(local()); } void clear() { v_.reset(); } }; } }
