// Common/Sptr.h -- smart-pointer typedef mixin exposed by the reference's public API
// (Common/Sptr.h:4-10: rt::RayTracer::sptr / ::uptr, used at OpenGLView/MainFrame.h:74).
#pragma once
#include <memory>

template <class T>
class ISptr {
public:
  using sptr = std::shared_ptr<T>;
  using uptr = std::unique_ptr<T>;
};
