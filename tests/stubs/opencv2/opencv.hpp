// Syntax-only stand-in (see tests/stubs/README.md): vo_node.cpp includes the umbrella header.
#include <opencv2/core.hpp>
