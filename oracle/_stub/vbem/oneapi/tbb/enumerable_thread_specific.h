// oracle/_stub/vbem — TEST INFRASTRUCTURE (see parallel_for.h): oneapi::tbb::enumerable_thread_specific with ONE thread's copy, made by the functor on first use
#pragma once
#include <functional>
#include <memory>
namespace oneapi { namespace tbb { template <class T> class enumerable_thread_specific { std::function<T()> make_; std::unique_ptr<T> v_; public: typedef T& reference;
  enumerable_thread_specific() : make_([] { return T(); }) {} template <class F> explicit enumerable_thread_specific(F f) : make_(f) {}
  T& local() { if (!v_) v_.reset(new T(make_())); return *v_; } }; } }
