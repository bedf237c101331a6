/*
 * Reproducer for "quirk (iv)" of DESIGN.md section 2, for a bifromq maintainer to confirm or refute against the real Java
 * (this repository has no JDK; the claim comes from the structural restatement in oracle/bmq_oracle.cpp).
 *
 * Claim: TenantRouteMatcher.matchAll (bifromq-dist-worker/.../cache/TenantRouteMatcher.java:88-156) does not terminate when
 *   - the publish topic ends with '/' (its last level is empty), e.g. "a/", and
 *   - the tenant holds more than 20 routes of a filter whose keys sort between the expansion filter "a" and the expansion
 *     filter "a/" -- e.g. 64 subscribers of the filter "a" with bucket bytes >= 0x01 -- so that the probe budget (20 next()
 *     calls, :127-131) runs out while the iterator still stands on "a".
 * Mechanism: the loop then seeks to tenantRouteStartKey(tenant, nextFilterLevels) with nextFilterLevels = ["a", ""] (:132-136),
 * i.e. to the byte string  ...a\\0\\0\\0 ; the keys of filter "a" are  ...a\\0\\0<bucket><flag>... : every key whose bucket byte
 * is >= 0x01 sorts AFTER that seek target, so itr.seek() lands on the very entry the loop is standing on, 20 probes are spent
 * on the same entries again, and so on.
 * Expected by the author of this file: the test below times out.  If it passes, the restatement is wrong at the seek -- please
 * report; the engine's results do not depend on it (the engine is checked against the semantic oracle, which has no such loop).
 *
 * Place next to TenantRouteMatcherTest (bifromq-dist-worker/src/test/java/org/apache/bifromq/dist/worker/cache/) -- it uses that
 * test's TreeMapKVReader fixture (TenantRouteMatcherTest.java:344-444) and helpers.
 */
package org.apache.bifromq.dist.worker.cache;

import static org.apache.bifromq.dist.worker.schema.KVSchemaUtil.toNormalRouteKey;
import static org.apache.bifromq.util.TopicUtil.from;
import static org.testng.Assert.assertEquals;

import com.google.protobuf.ByteString;
import java.util.Map;
import java.util.Set;
import java.util.TreeMap;
import org.apache.bifromq.basekv.utils.BoundaryUtil;
import org.apache.bifromq.util.BSUtil;
import org.testng.annotations.Test;

public class TenantRouteMatcherTrailingSlashTest extends TenantRouteMatcherTest {
    @Test(timeOut = 10_000) // a correct implementation answers in microseconds
    public void topicEndingWithSlashAgainstManyRoutesOfItsPrefixFilter() {
        String tenantId = "tenantA";
        TreeMap<ByteString, ByteString> kv = new TreeMap<>(BoundaryUtil::compare);
        for (int i = 0; i < 64; i++) { // 64 receivers: their bucket bytes cover >= 0x01 many times over
            String receiverUrl = toReceiverUrl(0, "inbox" + i, "deliverer" + (i % 7));
            kv.put(toNormalRouteKey(tenantId, from("a"), receiverUrl), BSUtil.toByteString(1L));
        }
        kv.put(toNormalRouteKey(tenantId, from("a/"), toReceiverUrl(0, "inboxX", "delivererX")), BSUtil.toByteString(1L));
        TenantRouteMatcher matcher = new TenantRouteMatcher(tenantId, () -> new TreeMapKVReader(kv), eventCollector, timer);
        Map<String, IMatchedRoutes> matched = matcher.matchAll(Set.of("a/"), 100, 100);
        assertEquals(matched.get("a/").routes().size(), 1); // only the filter "a/" matches the topic "a/"
    }
}
