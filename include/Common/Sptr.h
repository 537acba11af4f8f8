// Common/Sptr.h -- the smart-pointer alias mixin that rt::RayTracer's public API inherits
// (the caller writes rt::RayTracer::uptr, OpenGLView/MainFrame.h:74).  Only the two member
// type names are part of the contract; everything else about this header is ours.
#pragma once
#include <memory>
#include <utility>

template <class T>
struct ISptr {
  typedef std::shared_ptr<T> sptr;   // shared ownership handle of the deriving class
  typedef std::unique_ptr<T> uptr;   // sole ownership handle of the deriving class

  // convenience factories (additive; the reference spells std::make_unique at the call site)
  template <class... Args>
  static uptr MakeUnique(Args&&... args) { return uptr(new T(std::forward<Args>(args)...)); }
  template <class... Args>
  static sptr MakeShared(Args&&... args) { return std::make_shared<T>(std::forward<Args>(args)...); }

protected:
  ISptr() = default;
  ~ISptr() = default;
};
