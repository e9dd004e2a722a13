// <CORA/CORA_vis.h> of the reference: Matplot++ visualisation of the iterates (out of scope).
// Present so that sources including it still compile; it declares nothing.
#pragma once
