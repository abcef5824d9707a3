// TEST INFRASTRUCTURE ONLY — stand-in for <boost/noncopyable.hpp>
#pragma once
namespace boost { class noncopyable { protected: noncopyable() = default; ~noncopyable() = default; noncopyable(const noncopyable &) = delete; noncopyable &operator=(const noncopyable &) = delete; }; }
