// oracle/_stub/boost/config.hpp — TEST INFRASTRUCTURE.  Boost is not in this image.  The reference's
// include/salmon/internal/util/SalmonMath.hpp includes "boost/config.hpp" only for the optional BOOST_LIKELY / BOOST_UNLIKELY
// macros and defines them itself when they are missing, so an empty header is enough to compile that one file for the pin in
// oracle/ref_defaults_shim.cpp.  Nothing of Boost is restated here.
