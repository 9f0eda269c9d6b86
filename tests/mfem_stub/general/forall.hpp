// TEST INFRASTRUCTURE: stands where MFEM's general/forall.hpp is included by the reference's remhos_tools.hpp; the declarations of
// that header use nothing from it (tests/test_binding_compiles.py).
#pragma once
